// Host-side checks of nerf_pl_amd/csrc/mlp_layout.h (the maps the pack kernels and the MLP kernels share): compiled with g++ and
// run by tests/test_layout_host.py.  Exit code 0 = every check passed; failures are printed.
#include <cstdio>
#include <set>
#include <vector>

#include "mlp_layout.h"

using namespace nerfhip::mlp;

static int fails = 0;
#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            ++fails;                              \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);             \
            std::printf("\n");                    \
        }                                         \
    } while (0)

int main() {
    // ---- every input column of every weight matrix is multiplied by exactly one B-operand slot ----
    for (int L = 0; L < kNumLayers; ++L) {
        const Layer ly = kLayers[L];
        const int in_features = kParamIn[ly.param];
        std::vector<int> hits(in_features, 0);
        int pads = 0;
        for (int ks = 0; ks < layer_slabs(L); ++ks)
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 8; ++j) {
                    const int c = layer_in_col(L, ks, h, j);
                    if (c < 0) { ++pads; continue; }
                    CHECK(c < in_features, "layer %d: column %d out of range %d", L, c, in_features);
                    if (c < in_features) ++hits[c];
                }
        for (int c = 0; c < in_features; ++c) CHECK(hits[c] == 1, "layer %d: column %d used %d times", L, c, hits[c]);
        CHECK(pads == 16 * layer_slabs(L) - in_features, "layer %d: %d pads", L, pads);
    }
    // ---- chain_feature: the 16 slots of a slab are its 16 features ----
    for (int ks = 0; ks < 16; ++ks) {
        std::set<int> f;
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 8; ++j) f.insert(chain_feature(ks, h, j));
        CHECK((int)f.size() == 16 && *f.begin() == 16 * ks && *f.rbegin() == 16 * ks + 15, "chain_feature slab %d", ks);
    }
    // ---- slab_nat_h / slab_nat_j invert chain_feature within a slab ----
    for (int i = 0; i < 16; ++i) CHECK(chain_feature(0, slab_nat_h(i), slab_nat_j(i)) == i, "slab_nat %d", i);
    // ---- gate bits: the 32 values of a gate word occupy 32 distinct bits; the pair of a packed dword k sits at bit 15-k of
    //      the low (even slot) and high (odd slot) half-word ----
    for (int w = 0; w < 4; ++w) {
        std::set<int> bits;
        for (int v = 0; v < 32; ++v) {
            const int idx = 32 * w + v;
            CHECK(gate_word(idx) == w, "gate_word %d", idx);
            bits.insert(gate_bit(idx));
            CHECK(gate_bit(idx) == 16 * (v & 1) + 15 - (v >> 1), "gate_bit %d", idx);
        }
        CHECK((int)bits.size() == 32, "gate word %d: %d distinct bits", w, (int)bits.size());
    }
    // ---- the packed forward stream ----
    for (int prec = 0; prec < 2; ++prec) {
        int frag = 0;
        for (int L = 0; L < kNumLayers; ++L) {
            CHECK(layer_pieces(L, prec) == kLayers[L].nt * layer_slabs(L) * ppf(prec), "layer_pieces %d", L);
            if (L + 1 < kNumLayers)
                CHECK(layer_start(L + 1, prec) - layer_start(L, prec) ==
                          layer_pieces(L, prec) + (L + 1 == kLoopSecond ? bias_block_pieces(prec) : 0),
                      "layer_start %d prec %d", L, prec);
            frag += layer_pieces(L, prec);
        }
        CHECK(frag == (prec ? 1056 : 2112), "fragment pieces %d", frag);     // (11 layers: xyz_encoding_final is folded into the dir layer)
        CHECK(total_pieces(prec) == frag + bias_block_pieces(prec), "total pieces");
        CHECK(bias_block_start(prec) == layer_start(kLoopSecond - 1, prec) + layer_pieces(kLoopSecond - 1, prec), "bias block position");
        CHECK(bias_block_pieces(prec) % kChunkPieces == 0 && bias_block_pieces(prec) >= kNumLayers, "bias block size");
        CHECK(padded_pieces(prec) % kChunkPieces == 0 && padded_pieces(prec) >= total_pieces(prec), "padding");
        // the two looped layer triples: same shapes, same chunk phase, same ring slots
        for (int k = 0; k < kLoopLayers; ++k) {
            CHECK(layer_pieces(kLoopFirst + k, prec) == layer_pieces(kLoopSecond + k, prec), "triple shape %d", k);
            CHECK(kLayers[kLoopFirst + k].enc_slabs == 0 && kLayers[kLoopSecond + k].enc_slabs == 0, "triple layers take no encoding");
            const int a = layer_start(kLoopFirst + k, prec), b = layer_start(kLoopSecond + k, prec);
            CHECK(a % kChunkPieces == b % kChunkPieces, "chunk phase %d", k);
            CHECK((a / kChunkPieces) % kSlots == (b / kChunkPieces) % kSlots, "ring slot %d", k);
        }
        CHECK(chunks_upto_layer(kNumLayers, prec) * kChunkPieces == padded_pieces(prec), "chunk count");
    }
    // ---- saved-activation sections tile the block without gaps ----
    CHECK(kActEncX == 0 && kActEncD == kXyzSlabs && kActH0 == kXyzSlabs + kDirSlabs, "encoding sections");
    for (int l = 1; l <= 8; ++l) CHECK(act_h(l) == kActH0 + 16 * (l - 1), "act_h %d", l);
    CHECK(kActFeat == act_h(8) + 16 && kActT == kActFeat + 16 && kActSlabs == kActT + 8, "tail sections");
    CHECK(kActSlabs % 2 == 0 && kDySlabs % 2 == 0, "pairs");
    CHECK(f8_act_tile_bytes() == (kF8ActPairs + kMaskPieces + 1) * kPieceBytes, "f8 X tile");
    CHECK(f8_dy_tile_bytes() == (kF8DyPairs + 1) * kPieceBytes, "f8 dY tile");
    {   // section indices: one per section, dense
        std::set<int> xs, ds;
        for (int s = 0; s < kActSlabs; ++s) xs.insert(f8_x_section(s));
        for (int s = 0; s < kDySlabs; ++s) ds.insert(f8_dy_section(s));
        CHECK((int)xs.size() == 12 && *xs.rbegin() == 11, "x sections %d", (int)xs.size());
        CHECK((int)ds.size() == 12 && *ds.rbegin() == 11, "dy sections %d", (int)ds.size());
        for (int l = 1; l <= 8; ++l) {
            CHECK(f8_x_section(act_h(l)) == 1 + l && f8_x_section(act_h(l) + 15) == 1 + l, "x section of h%d", l);
            CHECK(f8_dy_section(dy_h(l)) == 4 + (8 - l) && f8_dy_section(dy_h(l) + 15) == 4 + (8 - l), "dy section of dY_%d", l);
        }
    }
    // ---- weight-gradient jobs: every parameter once, its X columns cover the input features exactly once.  The dir job's second X
    // section is h8, not the final layer's output f (round 6: f is not saved): its 256 columns are the rows of G, which
    // mlp_bwd_fold_kernel turns into dW_dir[:, 0..255] = G W_f^T + s b_f^T — they stand for those 256 input columns ----
    {
        std::set<int> params;
        for (int j = 0; j < kNumDwJobs; ++j) {
            const DwJob jb = kDwJobs[j];
            params.insert(jb.param);
            const int in_features = kParamIn[jb.param], out_features = kParamOut[jb.param];
            CHECK(jb.dy_slabs * 16 >= out_features && (jb.dy_slabs - 2) * 16 < out_features + 16, "job %d dY slabs", j);
            std::vector<int> hits(in_features, 0);
            for (int part = 0; part < 2; ++part) {
                const int slabs = part ? jb.x2_slabs : jb.x1_slabs, col0 = part ? jb.x2_col0 : jb.x1_col0, enc = part ? jb.x2_enc : jb.x1_enc;
                for (int ks = 0; ks < slabs; ++ks)
                    for (int h = 0; h < 2; ++h)
                        for (int jj = 0; jj < 8; ++jj) {
                            int c = (enc == 0 || enc == kDwEncFold) ? chain_feature(ks, h, jj) : enc == 1 ? xyz_slot_channel(ks, h, jj) : dir_slot_channel(ks, h, jj);
                            if (c < 0) continue;
                            c += col0;
                            if (c < in_features) ++hits[c];
                        }
            }
            for (int c = 0; c < in_features; ++c) CHECK(hits[c] == 1, "job %d (param %d): input column %d covered %d times", j, jb.param, c, hits[c]);
            CHECK((jb.x1_slabs + jb.x2_slabs) / 2 <= kDwMaxXTiles, "job %d X tiles", j);
        }
        CHECK((int)params.size() == 12, "jobs cover %d parameters", (int)params.size());
        CHECK(kDwJobs[kDwJobDir].x2_enc == kDwEncFold && kDwJobs[kDwJobDir].x2_off == act_h(8) && kDwJobs[kDwJobDir].x2_col0 == 0,
              "the dir job multiplies dY_dir by h8 (fold)");
        // no job reads the sections that are no longer written
        for (int j = 0; j < kNumDwJobs; ++j) {
            if (j == kDwJobFinal) continue;                    // derived: no workgroups
            const DwJob jb = kDwJobs[j];
            CHECK(jb.dy_off + jb.dy_slabs <= kDyFeat || jb.dy_off >= kDyFeat + 16, "job %d reads dL/d(final)", j);
            CHECK(jb.x1_off + jb.x1_slabs <= kActFeat || jb.x1_off >= kActFeat + 16, "job %d reads f (x1)", j);
            CHECK(jb.x2_slabs == 0 || jb.x2_off + jb.x2_slabs <= kActFeat || jb.x2_off >= kActFeat + 16, "job %d reads f (x2)", j);
        }
        CHECK(kFoldScratchFloats == 128 * 256 + 128 && kFoldPieces == 256 + 128 + 1, "fold scratch / image block sizes");
    }
    // ---- backward chain stream ----
    for (int prec = 0; prec < 2; ++prec) {
        int n = 0;
        for (int L = 0; L < kNumBwdLayers; ++L) {
            CHECK(bwd_layer_start(L, prec) == n, "bwd_layer_start %d", L);
            n += bwd_layer_pieces(L, prec);
        }
        CHECK(bwd_total_pieces(prec) == n && bwd_padded_pieces(prec) % kChunkPieces == 0, "bwd stream size");
    }
    // ---- saved-tensor block addressing (round 5): piece p of wave tile t -> tile_block_off(t) + p * il KiB is a bijection onto the
    // buffer for the plain (il = 1) and the interleaved (il = 8) layout, and il = 1 is the contiguous tile-major block ----
    for (int il : {1, 8}) {
        const int pieces = kActSlabs + kMaskPieces, tiles = 24, tile_bytes = pieces * kPieceBytes;
        std::set<size_t> seen;
        for (int t = 0; t < tiles; ++t)
            for (int pc = 0; pc < pieces; ++pc) {
                const size_t off = tile_block_off(t, tile_bytes, il) + (size_t)pc * il * kPieceBytes;
                CHECK(off % kPieceBytes == 0 && off + kPieceBytes <= (size_t)tiles * tile_bytes, "il %d: tile %d piece %d out of the buffer", il, t, pc);
                CHECK(seen.insert(off).second, "il %d: tile %d piece %d collides", il, t, pc);
                if (il == 1) CHECK(off == (size_t)t * tile_bytes + (size_t)pc * kPieceBytes, "il 1 is tile-major (tile %d piece %d)", t, pc);
                if (il == 8) CHECK(off / kPieceBytes % 8 == (size_t)(t % 8) && off / (8 * (size_t)tile_bytes) == (size_t)(t / 8), "il 8: octet / lane of tile %d piece %d", t, pc);
            }
        CHECK((int)seen.size() == tiles * pieces, "il %d covers the buffer", il);
    }
    CHECK(act_il(0) == 1 && act_il(1, true) == 1 && act_il(1) == NERFHIP_ACT_IL, "interleave applies to the bf16 slab blocks only");
    if (fails == 0) std::printf("layout ok\n");
    return fails ? 1 : 0;
}
