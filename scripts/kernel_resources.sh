# Compiler's view of every kernel of the final source (no GPU needed): registers, spills, LDS, code size, instruction mix.
# usage: bash scripts/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd $R/libmultirobotplanning_amd/csrc
for src in ll_kernel.hip conflict_kernel.hip; do
  echo "== $src  (hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I $R/include -c $src -o $T/k.o \
    -Rpass-analysis=kernel-resource-usage --save-temps=obj 2>&1 | grep -E "Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" |
    sed 's/^.*remark: [^ ]* *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - - - | sed 's/Function Name: //'
  S=$T/${src%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s
  echo "-- code bytes per kernel (llvm-readelf -s)"
  /opt/rocm/lib/llvm/bin/llvm-readelf -s $T/${src%.hip}-hip-amdgcn-amd-amdhsa-gfx950.out | awk '$4=="FUNC" && $8 ~ /mrp_ll/ {print $3, $8}' | sort -u -k2
  echo "-- static instruction mix per kernel: total / s_* / v_* / ds_* / global_* / flat_* / s_cbranch* / v_readlane+v_writelane"
  for k in $(grep -oE "^mrp_ll[a-z_]*:" $S | tr -d ':'); do
    awk "/^$k:/{f=1} f&&/^\.Lfunc_end/{f=0} f" $S > $T/one.s
    printf "%s %d / %d / %d / %d / %d / %d / %d / %d\n" $k \
      $(grep -cE "^\s+(s_|v_|ds_|global_|flat_|buffer_|scratch_)" $T/one.s) $(grep -cE "^\s+s_" $T/one.s) $(grep -cE "^\s+v_" $T/one.s) \
      $(grep -cE "^\s+ds_" $T/one.s) $(grep -cE "^\s+global_" $T/one.s) $(grep -cE "^\s+flat_" $T/one.s) \
      $(grep -cE "^\s+s_cbranch" $T/one.s) $(grep -cE "^\s+v_(readlane|writelane)" $T/one.s)
  done
done
rm -rf $T
