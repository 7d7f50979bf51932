#!/bin/bash
# Kernel experiments: build variants of the quad-layout 12-wave kernels (k_6_7.o) with extra -D flags and link one
# library per variant (juqbox.jl_amd/csrc/../exp/libjq_<tag>.so; selected with JQ_LIB=<path>).
# usage: scripts/exp_variants.sh tag1:"-DA -DB" tag2:"" ...
# (every variant library reports "<version> src:<hash>+<tag>" from jq_version(): profiles recorded for the production build cannot be
#  joined with a variant, and vice versa)
cd "$(dirname "$0")/../juqbox.jl_amd/csrc"
mkdir -p build/exp ../exp
# EXP_OBJ / EXP_VARIANT: which object is rebuilt (default: k_6_7.o = quad layout, 12 waves; EXP_OBJ=u_6_7 EXP_VARIANT=9: cooperative quad)
EXP_OBJ=${EXP_OBJ:-k_6_7}; EXP_VARIANT=${EXP_VARIANT:-0}
SCHED=${EXP_SCHED-$([ "$EXP_VARIANT" = 0 ] && echo "-mllvm -amdgpu-sched-strategy=iterative-maxocc")}
OBJS=$(ls build/*.o | grep -v "build/$EXP_OBJ.o")
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DJQ_NT=6 -DJQ_BW=7 -DJQ_VARIANT=$EXP_VARIANT $flags \
      -mllvm -amdgpu-mfma-vgpr-form=1 $SCHED -save-temps=obj -c jq_kernel_inst.hip -o build/exp/k_$tag.o 2>build/exp/k_$tag.log \
    && echo "const char jq_variant_tag[] = \"$tag\";" > build/exp/tag_$tag.c && gcc -fPIC -c build/exp/tag_$tag.c -o build/exp/tag_$tag.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libjq_$tag.so $OBJS build/exp/k_$tag.o build/exp/tag_$tag.o -ldl -pthread \
    && echo "$tag: $(grep -E '; ScratchSize|; NumVgprs' build/exp/jq_kernel_inst-hip-amdgcn-amd-amdhsa-gfx950.s 2>/dev/null | tr '\n' ' ')" ) &
  # (-save-temps files collide between parallel jobs: serialise)
  wait
done
ls -la ../exp/*.so | awk '{print $5, $9}'
