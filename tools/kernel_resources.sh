#!/bin/bash
# No GPU needed: registers, spills, LDS and code size of every gfx950 kernel of libnerfhip, from hipcc's own remarks
# (-Rpass-analysis=kernel-resource-usage) and the assembler's codeLenInByte.   tools/kernel_resources.sh > profiles/<file>.txt
cd "$(dirname "$0")/.." || exit 1
TMP=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Iinclude --cuda-device-only -S -Rpass-analysis=kernel-resource-usage"
digest() {   # $1 = label, rest = hipcc args
  label=$1; shift
  hipcc $FLAGS "$@" -o $TMP/k.s 2> $TMP/k.log
  python3 - "$label" $TMP/k.log $TMP/k.s <<'PY'
import re, sys
label, log, asm = sys.argv[1:]
txt = open(log).read()
sizes = dict(re.findall(r"^\s*\.size\s+(\S+), \.Lfunc_end\d+-\S+\n(?:.*\n)*?; codeLenInByte = (\d+)", open(asm).read(), flags=re.M))
for blk in txt.split("Function Name: ")[1:]:
    name = blk.split(" ")[0].strip()
    def g(k):
        m = re.search(k + r": (\d+)", blk)
        return m.group(1) if m else "?"
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "").replace("nerfhip::", "")
    print("%-16s %-58s VGPR %3s  AGPR %3s  SGPR %3s  spill v/s %s/%s  scratch %4s B  LDS %6s B  waves/SIMD %s  code %6s B"
          % (label, dem[:58], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
             g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]"), sizes.get(name, "?")))
PY
}
for P in 1 0; do for M in 1 0; do for V in 0 1 2 3; do
  [ "$P" = 0 ] && [ "$V" = 3 ] && continue
  EXTRA=""; [ "$V" = 3 ] && EXTRA="-mllvm -amdgpu-sched-strategy=max-memory-clause"
  digest "fwd p$P m$M v$V" -DNH_PREC=$P -DNH_MODE=$M -DNH_VARIANT=$V $EXTRA nerf_pl_amd/csrc/mlp_fwd_variant.hip
done; done; done
for P in 1 0; do for SV in 0 1 2; do
  [ "$P" = 0 ] && [ "$SV" = 2 ] && continue
  EXTRA=""; [ "$SV" = 2 ] && EXTRA="-mllvm -amdgpu-sched-strategy=max-memory-clause"
  digest "render p$P sv$SV" -DNH_PREC=$P -DNH_SV=$SV $EXTRA nerf_pl_amd/csrc/mlp_render_variant.hip
done; done
digest mlp_bwd_chain -mllvm -amdgpu-sched-strategy=max-memory-clause nerf_pl_amd/csrc/mlp_bwd_chain.hip
for f in mlp_bwd mlp_dx mlp_pack prologue draws sampling composite posenc loss optim rays linear; do digest $f nerf_pl_amd/csrc/$f.hip; done
rm -rf $TMP
