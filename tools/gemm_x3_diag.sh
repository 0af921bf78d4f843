# diagnostic builds of the fp16x3 contraction kernel for tools/gemm_x3_diag.py (never loaded by the package)
cd "$(dirname "$0")/../interactron_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -fno-slp-vectorize"
for v in NOCONV NOMMA NOLOAD HALFBAR "NOCONV -DX3_DIAG_NOMMA" "NOCONV -DX3_DIAG_NOLOAD" "NOCONV -DX3_DIAG_NOMMA -DX3_DIAG_NOLOAD"; do
  n=$(echo "$v" | sed -e 's/ -DX3_DIAG_/_/g')
  hipcc $F -DX3_DIAG_$v -x hip -c gemm.hip -o ../lib/obj/gemm_x3diag_$n.o &
done
wait
for v in NOCONV NOMMA NOLOAD HALFBAR NOCONV_NOMMA NOCONV_NOLOAD NOCONV_NOMMA_NOLOAD; do
  hipcc -shared -fPIC --offload-arch=gfx950 $(ls ../lib/obj/*.hip.o ../lib/obj/*.cpp.o | grep -v "gemm.hip.o") ../lib/obj/gemm_x3diag_$v.o -o ../lib/libx3diag_$v.so
done
ls ../lib/libx3diag_*.so
