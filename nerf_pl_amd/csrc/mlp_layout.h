// Packed-weight stream layout of the fused NeRF MLP (reference models/nerf.py:42-124, default
// architecture D=8, W=256, skips=[4], 63/27 input channels).
//
// The kernel keeps the activations of 32 points per wavefront IN REGISTERS for the whole network:
// the MFMA computes   H_out^T[256 x 32pts] = W[256 x K] * H_in^T[K x 32pts]   with the weights as the
// A operand and the points as the N dimension, so the C/D fragment of one layer (lane = point,
// registers = output features) is — after bias/ReLU/(bf16 pack) — directly the B operand of the next
// layer.  No cross-lane traffic is needed because we choose which 8 input features a lane's B
// registers mean and permute the COLUMNS of W to match when packing:
//     C/D layout of v_mfma_f32_32x32x*:  lane l -> col n = l&31, half h = l>>5;
//                                        reg r  -> row (r&3) + 8*(r>>2) + 4*h     (r in 0..15)
//     B "slab" ks (16 features) of a 32-row tile t = ks/2, s = ks%2 uses regs 8s..8s+7, i.e.
//     slot (ks,h,j) <-> feature 16*ks + 8*(j>>2) + 4*h + (j&3).
// Weights stream through LDS as 1 KiB "pieces" (64 lanes x 16 B, exactly one ds_read_b128 per wave)
// in the precise order the kernel consumes them, so both the global->LDS DMA and the LDS reads are
// lane-linear and bank-conflict free.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NH_HD __host__ __device__
#else
#define NH_HD
#endif

namespace nerfhip {

// Coarse depths formed in the prologue of the fused forward (MODE_RAYS; rendering.py:183-204): when z_out is set the kernel
// computes z of its points itself from the ray bounds (+ the caller's jitter draw `prand` when perturb > 0), uses them and
// writes them to z_out (B,S) for the compositing that follows — instead of reading a z tensor an earlier launch produced.
struct FwdZGen {
    const float* prand;
    float* z_out;
    int use_disp;
    float perturb;
    // activation-saving bf16 forward on rays: 1 = do NOT store the 6 input-encoding slabs of a tile block (mlp_layout.h kActEncX /
    // kActEncD) — the weight-gradient launch regenerates them from the rays and the depths (nerfhip_mlp_bwd_multi_rays)
    int skip_enc_save = 0;
};

namespace mlp {

constexpr int kPieceBytes = 1024;
constexpr int kChunkPieces = 32;                       // 32 KiB per ring slot
constexpr int kChunkBytes = kPieceBytes * kChunkPieces;
constexpr int kSlots = 3;                              // LDS ring: 96 KiB

constexpr int kXyzCh = 63, kDirCh = 27, kW = 256;
constexpr int kXyzSlabs = 4;   // 63 channels (+1 pad) = 64 slots
constexpr int kDirSlabs = 2;   // 27 channels (+5 pad) = 32 slots

enum InKind { IN_XYZ = 0, IN_CHAIN = 1, IN_XYZ_CHAIN = 2, IN_DIR_CHAIN = 3 };

struct Layer {
    int param;        // index in state_dict order: 0..7 xyz_encoding_1..8, 8 final, 9 dir, 10 sigma, 11 rgb
    int nt;           // 32-row output tiles
    int n_out;        // real output features
    int kind;         // InKind
    int enc_slabs;    // leading slabs fed by an input encoding (xyz or dir)
    int chain_slabs;  // slabs fed by the previous layer's registers
};

// kernel execution order.  xyz_encoding_final is NOT a layer of the kernels (round 6): it has no activation (nerf.py:70,116), so
//     dir_encoding(cat([final(h8), dir])) = relu(W_c h8 + W_dd enc_d + b_c),   W_c = W_dir[:, :256] W_final,  b_c = W_dir[:, :256] b_final + b_dir
// and the dir layer runs on h8 directly with the PRODUCT matrix, which the pack kernels form in fp32 (fold_wc / fold_bc in
// mlp_pack_pieces.h: one 256-term fmaf chain per element, the same chain for the forward image and the W^T image, so both round to
// the same bf16).  11 % fewer MFMAs per point in the forward, 18 % fewer in the backward chain; the parameter gradients of both
// original layers follow from G = dL/dW_c (kDwJobs below).  The layer-by-layer path (linear.hip) keeps the reference's two layers.
constexpr int kNumLayers = 11;
constexpr int kSigmaLayer = 8;     // index (in this table) of the sigma head
constexpr int kDirLayer = 9;       // ... of the dir layer (folded: weights W_c, bias b_c)
constexpr Layer kLayers[kNumLayers] = {
    {0, 8, 256, IN_XYZ, 4, 0},         // xyz_encoding_1          nerf.py:62-63
    {1, 8, 256, IN_CHAIN, 0, 16},      // xyz_encoding_2
    {2, 8, 256, IN_CHAIN, 0, 16},      // xyz_encoding_3
    {3, 8, 256, IN_CHAIN, 0, 16},      // xyz_encoding_4
    {4, 8, 256, IN_XYZ_CHAIN, 4, 16},  // xyz_encoding_5 (skip: cat([input_xyz, h]))   nerf.py:64-65,108-109
    {5, 8, 256, IN_CHAIN, 0, 16},      // xyz_encoding_6
    {6, 8, 256, IN_CHAIN, 0, 16},      // xyz_encoding_7
    {7, 8, 256, IN_CHAIN, 0, 16},      // xyz_encoding_8
    {10, 1, 1, IN_CHAIN, 0, 16},       // sigma (raw)                                   nerf.py:78,112
    {9, 4, 128, IN_DIR_CHAIN, 2, 16},  // dir_encoding o xyz_encoding_final on [input_dir | h8]   nerf.py:70,73-75,116-118
    {11, 1, 3, IN_CHAIN, 0, 8},        // rgb (sigmoid)                                 nerf.py:79-81
};

// reference in_features per state_dict entry (row stride of weight matrices)
constexpr int kParamIn[12] = {63, 256, 256, 256, 319, 256, 256, 256, 256, 283, 256, 128};
constexpr int kParamOut[12] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 128, 1, 3};

NH_HD constexpr int layer_slabs(int L) { return kLayers[L].enc_slabs + kLayers[L].chain_slabs; }
// pieces per (slab, tile) fragment: bf16 = 1 (8 x bf16 per lane), fp32 = 2 (8 x f32 per lane)
NH_HD constexpr int ppf(int prec) { return prec == 0 /*NERFHIP_F32*/ ? 2 : 1; }
// ---- the packed stream: per layer its nt * slabs A fragments (execution order); the 12 bias pieces (256 fp32 each, piece L =
// bias of layer L) form one block between layers 4 and 5, padded to whole chunks.  The kernels copy the bias block into an LDS
// image once and never read it from the ring.  Consequences the kernels rely on (static_asserts in mlp_fwd_kernel.h):
//   * the 256 -> 256 layers are 128 (fp32: 256) pieces = whole chunks, so layers 1-3 and 5-7 see the chunk boundaries at the
//     same steps;
//   * the pad makes layer 5 start a multiple of kSlots chunks after layer 1, so both triples also use the same ring slots:
//     one copy of the code of "three 256 -> 256 layers" serves both (a runtime loop; the unrolled network would not fit the
//     64 KiB instruction cache).
NH_HD constexpr int layer_pieces(int L, int prec) { return layer_slabs(L) * kLayers[L].nt * ppf(prec); }
constexpr int kLoopFirst = 1, kLoopSecond = 5, kLoopLayers = 3;    // layers [1,4) and [5,8) share their code
NH_HD constexpr int raw_start(int L, int prec) {                   // without the bias block
    int g = 0;
    for (int i = 0; i < L; ++i) g += layer_pieces(i, prec);
    return g;
}
NH_HD constexpr int bias_block_pieces(int prec) {                  // >= kNumLayers pieces, whole chunks, slot-aligning
    int chunks = (kNumLayers + kChunkPieces - 1) / kChunkPieces;
    while (((raw_start(kLoopSecond, prec) - raw_start(kLoopFirst, prec)) / kChunkPieces + chunks) % kSlots != 0) ++chunks;
    return chunks * kChunkPieces;
}
NH_HD constexpr int bias_block_start(int prec) { return raw_start(kLoopSecond, prec); }
NH_HD constexpr int layer_start(int L, int prec) { return raw_start(L, prec) + (L >= kLoopSecond ? bias_block_pieces(prec) : 0); }
NH_HD constexpr int total_pieces(int prec) { return layer_start(kNumLayers, prec); }
NH_HD constexpr int padded_pieces(int prec) {
    return (total_pieces(prec) + kChunkPieces - 1) / kChunkPieces * kChunkPieces;
}
NH_HD constexpr int chunks_upto_layer(int Lend, int prec) {   // chunks needed to run layers [0, Lend)
    return (layer_start(Lend, prec) + kChunkPieces - 1) / kChunkPieces;
}
// chunks between the two looped layer triples (a multiple of kSlots)
NH_HD constexpr int loop_chunk_shift(int prec) { return (layer_start(kLoopSecond, prec) - layer_start(kLoopFirst, prec)) / kChunkPieces; }
static_assert(layer_start(kLoopFirst, 0) % kChunkPieces == 0 && layer_start(kLoopFirst, 1) % kChunkPieces == 0 &&
                  layer_start(kLoopSecond, 0) % kChunkPieces == 0 && layer_start(kLoopSecond, 1) % kChunkPieces == 0,
              "the looped layer triples start on chunk boundaries");
static_assert(loop_chunk_shift(0) % kSlots == 0 && loop_chunk_shift(1) % kSlots == 0, "... in the same ring slots");
static_assert(bias_block_pieces(0) >= kNumLayers && bias_block_pieces(1) >= kNumLayers, "the bias block holds every layer's bias");
static_assert(layer_pieces(kLoopFirst, 1) == layer_pieces(kLoopSecond, 1) && layer_pieces(kLoopFirst + 2, 1) == layer_pieces(kLoopSecond + 2, 1),
              "the triples are layer-for-layer the same shape");

// ---- fragment order inside a layer: output-tile-major, fragment i = (tile i / nks, slab i % nks) -----------------------
// (pack kernel and MLP kernels share these two functions)
NH_HD constexpr int frag_tile(int i, int nt, int nks) { return i / nks; }
NH_HD constexpr int frag_slab(int i, int nt, int nks) { return i % nks; }

// ---- input-slot maps --------------------------------------------------------------------------
// Encoding channel (reference order, nerf.py:33-38: [x, sin f0 x, cos f0 x, sin f1 x, ...]) held by
// slot (ks,h,j) of an encoding with F frequencies spread over `slabs` slabs; -1 = zero padding.
// Half h owns frequencies k = 2i+h, so one sincos per lane yields both channels it needs;
// the 3 identity channels fill the tail (h=0: x,y ; h=1: z).
NH_HD constexpr int enc_slot_channel(int F, int slabs, int ks, int h, int j) {
    const int idx = 8 * ks + j;                // 0 .. 8*slabs-1 within this half
    const int npair = 3 * (F / 2);             // (i,c) pairs per half
    if (idx < 2 * npair) {
        const int p = idx >> 1, i = p / 3, c = p % 3, k = 2 * i + h;
        return 3 + 6 * k + c + 3 * (idx & 1);
    }
    const int tail = idx - 2 * npair;          // identity channels
    if (h == 0) return tail < 2 ? tail : -1;
    return tail == 0 ? 2 : -1;
}
NH_HD constexpr int xyz_slot_channel(int ks, int h, int j) { return enc_slot_channel(10, kXyzSlabs, ks, h, j); }
NH_HD constexpr int dir_slot_channel(int ks, int h, int j) { return enc_slot_channel(4, kDirSlabs, ks, h, j); }
NH_HD constexpr int chain_feature(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }

// Column of the reference weight matrix W[out][in] multiplied by slot (ks,h,j) of layer L; -1 = pad.  (Dir layer: columns 0..255 are
// those of the product matrix W_c — h8 features — in the place of W_dir's, the final layer's outputs.)
NH_HD constexpr int layer_in_col(int L, int ks, int h, int j) {
    const Layer& ly = kLayers[L];
    if (ks < ly.enc_slabs) {
        if (ly.kind == IN_DIR_CHAIN) {
            const int c = dir_slot_channel(ks, h, j);
            return c < 0 ? -1 : kW + c;                       // cat([final(256), dir(27)])  nerf.py:118
        }
        return xyz_slot_channel(ks, h, j);                    // xyz first in cat([xyz, h])   nerf.py:109
    }
    const int f = chain_feature(ks - ly.enc_slabs, h, j);
    return ly.kind == IN_XYZ_CHAIN ? kXyzCh + f : f;
}

// ================================================================================================
// Training: saved-activation block, backward chain stream, weight-gradient jobs
// ================================================================================================
// Forward (SAVE variant) stores, per 32-point wave tile, every B-operand slab it builds, exactly as
// held in registers: slab s of tile T lives at  ((T*kActSlabs + s)*64 + lane) * sizeof(Slab)
// (bf16: 16 B/lane = one 1 KiB piece; fp32: 32 B/lane).  4.94 KB/point in bf16.
constexpr int kActEncX = 0;                 // 4 slabs  xyz encoding (slot order)
constexpr int kActEncD = 4;                 // 2 slabs  dir encoding
constexpr int kActH0 = 6;                   // h1..h8: 8 x 16 slabs (post-ReLU)
NH_HD constexpr int act_h(int l) { return kActH0 + 16 * (l - 1); }   // l = 1..8
constexpr int kActFeat = kActH0 + 128;      // 16 slabs  xyz_encoding_final output (no activation): NOT saved since round 6 (see kDwJobs); the slots stay
constexpr int kActT = kActFeat + 16;        // 8 slabs   dir_encoding output (post-ReLU)
constexpr int kActSlabs = kActT + 8;        // 158
// ReLU gates: after the slabs of a tile, one 1 KiB piece per gated layer (h1..h8, t): a lane's 16 B are four words, word w
// covering output tiles 2w, 2w+1 = the 32 values idx = 8*ks + j (slab ks = 4w .. 4w+3, slot j) the lane holds of them, i.e. the
// 16 packed dwords k = (idx & 31) >> 1 the forward produces in order; [pre-activation > 0] of value idx is bit gate_bit(idx) of
// word gate_word(idx): the even slots fill the low half-word, the odd slots the high one, first dword at the top (the forward
// shifts the pair of gate bits of each packed bf16 dword in with two packed 16-bit operations).  The backward chain reads these
// 9 KiB per tile instead of the 136 KiB of activation slabs.
NH_HD constexpr int gate_word(int idx) { return idx >> 5; }
NH_HD constexpr int gate_bit(int idx) { return 16 * (idx & 1) + 15 - ((idx & 31) >> 1); }
constexpr int kMaskPieces = 9;
NH_HD constexpr int mask_piece_h(int l) { return l - 1; }            // l = 1..8
constexpr int kMaskPieceT = 8;
NH_HD constexpr int slab_bytes(int prec) { return prec == 0 /*NERFHIP_F32*/ ? 32 * 64 : 16 * 64; }
NH_HD constexpr int act_mask_off(int prec) { return kActSlabs * slab_bytes(prec); }
NH_HD constexpr int act_tile_bytes(int prec) { return act_mask_off(prec) + kMaskPieces * kPieceBytes; }

// Optional interleave of the saved-tensor blocks (bf16 slabs; round 5 experiment, OFF by default).  A wave tile's block (activations:
// 158 slabs + 9 gate pieces; dY: 156 slabs) is one contiguous run of 1 KiB pieces.  The 8 waves of a workgroup store their slab s at
// the same time, 8 pieces that lie 156-167 KiB apart; with NERFHIP_ACT_IL = 8 the blocks of 8 consecutive tiles (one workgroup's) are
// interleaved piece by piece,
//     piece p of tile t  ->  byte ((t / IL) * IL * pieces_per_tile + p * IL + t % IL) * 1 KiB,
// so that they form ONE 8 KiB run.  A pure-store kernel of the chain's shape streams 5.67 TB/s in that layout against 5.10 TB/s
// tile-major (tools/probes/write_layout.hip), but the real kernels do not notice: same-box A/B of the whole step 1.0621 / 1.0674 /
// 1.0627 ms interleaved against 1.0636 / 1.0666 / 1.0654 ms tile-major, chain 279-286 us either way (profiles/r05_ab_interleave.txt;
// all 111 GPU tests of the paths that write or read the blocks pass with IL = 8).  Every producer and consumer addresses the blocks
// through act_il() / tile_block_off(), so the switch stays as a build flag.  fp32 slabs and the e4m3 pair pieces are always IL = 1.
#ifndef NERFHIP_ACT_IL
#define NERFHIP_ACT_IL 1
#endif
// The saved-tensor buffers are sized in whole bf16 workgroups = 8 wave tiles (nerfhip_mlp_act_bytes / _dy_bytes): an interleave group
// must divide that, or the last group of a buffer would be addressed past its end.
static_assert(NERFHIP_ACT_IL == 1 || NERFHIP_ACT_IL == 2 || NERFHIP_ACT_IL == 4 || NERFHIP_ACT_IL == 8,
              "NERFHIP_ACT_IL must divide the 8 wave tiles the bf16 buffers are padded to");
NH_HD constexpr int act_il(int prec, bool f8 = false) { return (prec == 1 /*NERFHIP_BF16*/ && !f8) ? NERFHIP_ACT_IL : 1; }
// byte offset of piece 0 of wave tile `tile` in a buffer of blocks of `tile_bytes`; its piece p follows at p * il KiB
NH_HD inline size_t tile_block_off(long long tile, int tile_bytes, int il) {
    return (size_t)(tile / il) * (size_t)il * (size_t)tile_bytes + (size_t)(tile % il) * kPieceBytes;
}

// Backward chain writes dL/d(pre-activation) slabs in the same format.
constexpr int kDyRgb = 0;                   // 2 slabs (3 real features, rest zero)
constexpr int kDyDir = 2;                   // 8 slabs
constexpr int kDyFeat = 10;                 // 16 slabs: dL/d(final) is NOT stored since round 6 (see kDwJobs); the slots stay
constexpr int kDySigma = 26;                // 2 slabs (1 real feature)
constexpr int kDyH0 = 28;                   // dY_8 .. dY_1: 8 x 16 slabs
NH_HD constexpr int dy_h(int l) { return kDyH0 + 16 * (8 - l); }     // l = 1..8
constexpr int kDySlabs = kDyH0 + 128;       // 156

// ---- backward chain: g_in = W^T g_out, W^T streamed as A operand ---------------------------------
struct BwdLayer {
    int param;      // W of this state_dict entry
    int nt;         // output tiles (input features of W / 32)
    int nks;        // slabs of g_out consumed
    int col0;       // first column of W used as output feature 0 (63 for the skip layer)
    int sigma_slab; // 1: the last slab is the sigma-head slab (W_sigma row 0 at slot h=0,j=0)
};
constexpr int kNumBwdLayers = 9;
constexpr int kBwdLayerFold = 1;           // index (in the table below) of the folded layer: its W^T fragments are those of W_c (kLayers above)
constexpr BwdLayer kBwdLayers[kNumBwdLayers] = {
    {11, 4, 1, 0, 0},    // rgb^T      : g_a_rgb(3)   -> g_t(128)
    {9, 8, 9, 0, 1},     // W_c^T+sigma^T : [g_a_dir(128) | g_sigma(1)] -> g_h8(256)   (W_c = W_dir[:, :256] W_final: dL/d(final) never exists)
    {7, 8, 16, 0, 0},    // L8^T : g_a8 -> g_h7
    {6, 8, 16, 0, 0},    // L7^T
    {5, 8, 16, 0, 0},    // L6^T
    {4, 8, 16, 63, 0},   // L5^T : hidden columns 63..318 of W_5 (skip)   nerf.py:109
    {3, 8, 16, 0, 0},    // L4^T
    {2, 8, 16, 0, 0},    // L3^T
    {1, 8, 16, 0, 0},    // L2^T : g_a2 -> g_h1
};
// Fragment order inside a backward-chain layer: output-tile-major (as the forward), fragment f = (tile f / nks, slab f % nks),
// so that a tile's epilogue (gate, pack, store) overlaps the next tile's MFMAs.
NH_HD constexpr int bwd_frag_tile(int f, int nt, int nks) { return f / nks; }
NH_HD constexpr int bwd_frag_slab(int f, int nt, int nks) { return f % nks; }
NH_HD constexpr int bwd_layer_pieces(int L, int prec) { return kBwdLayers[L].nks * kBwdLayers[L].nt * ppf(prec); }
NH_HD constexpr int bwd_layer_start(int L, int prec) {
    int g = 0;
    for (int i = 0; i < L; ++i) g += bwd_layer_pieces(i, prec);
    return g;
}
NH_HD constexpr int bwd_total_pieces(int prec) { return bwd_layer_start(kNumBwdLayers, prec); }
NH_HD constexpr int bwd_padded_pieces(int prec) {
    return (bwd_total_pieces(prec) + kChunkPieces - 1) / kChunkPieces * kChunkPieces;
}
NH_HD constexpr int bwd_chunks(int prec) { return bwd_padded_pieces(prec) / kChunkPieces; }

// ---- weight-gradient jobs: dW[o][i] = sum_p dY[p][o] * X[p][i] ------------------------------------
// One job = one (dY section, X sections) pair; a workgroup's wave w owns output tile w (32 dY features)
// against all X tiles.  X = [x1 | x2] slabs (x2 may be empty); columns map back to the reference
// weight matrix through (x?_col0, x?_enc): enc 0 = chain features (natural), 1 = xyz slots, 2 = dir slots,
// 3 = chain features of the fold scratch (below).
//
// The linear layer folded out of the saved tensors (round 6).  xyz_encoding_final has NO activation (nerf.py:70,116):
//     f = W_f h8 + b_f,   u = W_dx f + W_dd enc_d + b_d   (dir_encoding's pre-activation, W_dir = [W_dx | W_dd], nerf.py:118)
// so with  G = sum_p dY_dir[p] h8[p]^T  (128 x 256, = dL/dW_c)  and  s = sum_p dY_dir[p] = db_dir  (128):
//     dW_dx = sum_p dY_dir f^T = G W_f^T + s b_f^T        dW_f = sum_p (W_dx^T dY_dir) h8^T = W_dx^T G        db_f = W_dx^T s
// Neither f (16 slabs of X per tile) nor dL/df (16 slabs of dY) exists anywhere: the forward and the chain run the folded layer
// (kLayers / kBwdLayers above), the dir job multiplies dY_dir by h8 — the job class it always had — and mlp_bwd_fold_kernel
// finishes the three gradients from G, s and an fp32 snapshot of W_f, W_dx, b_f (the fold block at the end of the packed W^T
// image).  10 % fewer saved bytes written and read per point, 11 % fewer dW FLOPs.
// Job kDwJobFinal keeps its table entry (parameter mapping) but has no workgroups and nothing in the reduce kernel.
struct DwJob {
    int param;
    int dy_off, dy_slabs;       // section in the dY block (slabs)
    int x1_off, x1_slabs, x1_col0, x1_enc;
    int x2_off, x2_slabs, x2_col0, x2_enc;
};
constexpr int kNumDwJobs = 12;
constexpr int kDwJobFinal = 8, kDwJobDir = 9, kDwJobSigma = 10;      // indices in kDwJobs
constexpr int kDwEncFold = 3;
constexpr DwJob kDwJobs[kNumDwJobs] = {
    {0, dy_h(1), 16, kActEncX, 4, 0, 1, 0, 0, 0, 0},                 // xyz_encoding_1 : X = enc_xyz
    {1, dy_h(2), 16, act_h(1), 16, 0, 0, 0, 0, 0, 0},
    {2, dy_h(3), 16, act_h(2), 16, 0, 0, 0, 0, 0, 0},
    {3, dy_h(4), 16, act_h(3), 16, 0, 0, 0, 0, 0, 0},
    {4, dy_h(5), 16, kActEncX, 4, 0, 1, act_h(4), 16, 63, 0},        // skip: [enc_xyz | h4]
    {5, dy_h(6), 16, act_h(5), 16, 0, 0, 0, 0, 0, 0},
    {6, dy_h(7), 16, act_h(6), 16, 0, 0, 0, 0, 0, 0},
    {7, dy_h(8), 16, act_h(7), 16, 0, 0, 0, 0, 0, 0},
    {8, kDyFeat, 16, act_h(8), 16, 0, 0, 0, 0, 0, 0},                // xyz_encoding_final: DERIVED (mlp_bwd_fold_kernel), no workgroups
    {9, kDyDir, 8, kActEncD, 2, 256, 2, act_h(8), 16, 0, kDwEncFold},   // dir_encoding: [enc_dir | h8] -> [dW_dd | G]
    {10, kDySigma, 2, act_h(8), 16, 0, 0, 0, 0, 0, 0},               // sigma
    {11, kDyRgb, 2, kActT, 8, 0, 0, 0, 0, 0, 0},                     // rgb
};
// fold scratch of one model (fp32, behind the launch's partial slabs in the dW workspace): G[128][256] then s[128]
constexpr int kFoldG = 0, kFoldS = 128 * kW;
constexpr int kFoldScratchFloats = kFoldS + 128;
// fold block of the packed W^T image (fp32 pieces of 256 floats behind bwd_padded_pieces): W_f rows, W_dx rows, b_f
constexpr int kFoldWf = 0;                  // 256 pieces: piece m = W_f[m][0..255]
constexpr int kFoldWdx = 256;               // 128 pieces: piece j = W_dir[j][0..255]
constexpr int kFoldBf = 384;                // 1 piece:    b_f[0..255]
constexpr int kFoldPieces = 385;
NH_HD constexpr int bwd_image_pieces(int prec) { return bwd_padded_pieces(prec) + kFoldPieces; }
constexpr int kDwMaxXTiles = 10;            // (4+16)/2
// fp32 partial-sum slab of one (job, split): [8 o-tiles][10 x-tiles][64 lanes][16] weights + [8][64] bias
constexpr int kDwSlabFloats = 8 * kDwMaxXTiles * 64 * 16 + 8 * 64;

// ================================================================================================
// fp8 storage of the saved tensors (dtype NERFHIP_BF16_F8): bf16 MFMA chain, e4m3 dW operands
// ================================================================================================
// The dW GEMM (points = K) is the only consumer of the saved activations X and of dY, and it is bound by the bytes it
// reads.  In this mode the activation-saving forward and the backward chain store every slab PAIR (2t, 2t+1) — the 32
// features of one MFMA operand tile, 32 points — as ONE 1 KiB piece of OCP e4m3 bytes, lane (n, h) holding
// [slab 2t: 8 B | slab 2t+1: 8 B] — the operand format of v_mfma_scale_f32_32x32x64_f8f6f4, whose 32-wide K blocks are
// exactly the 32 points of a wave tile:
// OCP e4m3 with one e8m0 scale byte per (wave tile, 16-slab section = one layer's activations or dY):
//      stored q = x / 2^(E-127), E = max(Emax - 7, 1), Emax = biased exponent of the block's max |x|
//      => |q| < 2^8 <= 448 (e4m3 max; v_cvt_scalef32_pk_fp8 does not saturate, it produces NaN above 464).
// The input encodings (|sin|, |cos| <= 1, coordinates << 448) use the fixed scale 2^0.
// Tile block of X:  [79 pair pieces][9 gate pieces (as before)][1 KiB: scale dwords, one per SECTION at f8_x_section()]
// Tile block of dY: [78 pair pieces][1 KiB: scale dwords, one per SECTION at f8_dy_section()]
constexpr int kF8ActPairs = kActSlabs / 2;          // 79
constexpr int kF8DyPairs = kDySlabs / 2;            // 78
NH_HD constexpr int f8_act_gate_off() { return kF8ActPairs * kPieceBytes; }
NH_HD constexpr int f8_act_scale_off() { return f8_act_gate_off() + kMaskPieces * kPieceBytes; }
NH_HD constexpr int f8_act_tile_bytes() { return f8_act_scale_off() + kPieceBytes; }                 // 89 KiB (bf16: 167)
NH_HD constexpr int f8_dy_scale_off() { return kF8DyPairs * kPieceBytes; }
NH_HD constexpr int f8_dy_tile_bytes() { return f8_dy_scale_off() + kPieceBytes; }                   // 79 KiB (bf16: 156)
// Scale tables: one DWORD (the e8m0 byte, zero-extended) per SECTION:
// section index of an activation slab: 0 encx, 1 encd, 2..9 h1..h8, 10 feat, 11 t
NH_HD constexpr int f8_x_section(int slab) {
    return slab < kActEncD ? 0 : (slab < kActH0 ? 1 : (slab < kActFeat ? 2 + (slab - kActH0) / 16 : (slab < kActT ? 10 : 11)));
}
// section index of a dY slab: 0 rgb, 1 dir, 2 feat, 3 sigma, 4 + (8 - l) dY_l
NH_HD constexpr int f8_dy_section(int slab) {
    return slab < kDyDir ? 0 : (slab < kDyFeat ? 1 : (slab < kDySigma ? 2 : (slab < kDyH0 ? 3 : 4 + (slab - kDyH0) / 16)));
}
// Operand row m (0..31) of a pair, as ds_read_b64_tr_b8 delivers it to the MFMA: slab 2t + (m >> 4), half (m >> 3) & 1,
// slot m & 7  (natural order swaps bits 2 and 3 of the in-slab index).
NH_HD constexpr int f8_row_h(int m) { return (m >> 3) & 1; }
NH_HD constexpr int f8_row_j(int m) { return m & 7; }

// natural feature index i (0..15) inside a slab  <->  slot (h, j)   (chain_feature with ks = 0)
NH_HD constexpr int slab_nat_h(int i) { return (i >> 2) & 1; }
NH_HD constexpr int slab_nat_j(int i) { return (i & 3) + 4 * (i >> 3); }

}  // namespace mlp
}  // namespace nerfhip
