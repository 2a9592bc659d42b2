#!/bin/bash
# VGPR / SGPR / scratch / LDS of every kernel in a built object (default: all of csrc), from the code object's metadata notes.
#   tools/kernel_regs.sh [capsaicin_amd/csrc/kernels.o ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LLVM=/opt/rocm/lib/llvm/bin
objs=("$@")
[ ${#objs[@]} -eq 0 ] && objs=("$ROOT"/capsaicin_amd/csrc/{kernels,trace8,bvh,ploc,post,context}.o)
tmp=$(mktemp -d)
for o in "${objs[@]}"; do
    $LLVM/llvm-objcopy --dump-section .hip_fatbin="$tmp/fat.bin" "$o" 2>/dev/null || continue
    $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat.bin" --output="$tmp/dev.co" --unbundle 2>/dev/null || continue
    $LLVM/llvm-readelf --notes "$tmp/dev.co" | awk -v f="$(basename "$o")" '
        /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {l=$2}
        /\.wavefront_size:/ {printf "%-14s vgpr %3d sgpr %3d scratch %4d lds %6d  %s\n", f, v, s, p, l, name}' | c++filt | sort -k10
done
rm -rf "$tmp"
