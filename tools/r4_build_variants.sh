# variant builds of the contraction kernels for same-box A/B runs (tools/r4_ab_libs.sh): lib/libab_<name>.so
#   sh tools/r4_build_variants.sh "pipe3:-DX3_PIPE3" "pipe3m16:-DX3_PIPE3 -DX3_M16"
cd "$(dirname "$0")/../interactron_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -fno-slp-vectorize"
for v in "$@"; do
  n=${v%%:*}; d=${v#*:}
  hipcc $F $d -x hip -c gemm.hip -o ../lib/obj/gemm_ab_$n.o &
done
wait
for v in "$@"; do
  n=${v%%:*}
  hipcc -shared -fPIC --offload-arch=gfx950 $(ls ../lib/obj/*.hip.o ../lib/obj/*.cpp.o | grep -v "gemm.hip.o") ../lib/obj/gemm_ab_$n.o -o ../lib/libab_$n.so
done
ls -la ../lib/libab_*.so
