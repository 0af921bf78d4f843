# A/B of two builds of the kernel library on the contraction shapes (alternating, same box):
#   sh tools/r4_ab_libs.sh interactron_amd/lib/libab_old.so      (B = the in-tree product library)
mkdir -p gpurun_out/r5ab
for i in 1 2; do
  IX_LIB_PATH=$PWD/$1 python tools/w256_bench.py > gpurun_out/r5ab/old_$i.txt 2>&1
  python tools/w256_bench.py > gpurun_out/r5ab/new_$i.txt 2>&1
done
paste <(cut -c1-60 gpurun_out/r5ab/old_2.txt) <(cut -c31-60 gpurun_out/r5ab/new_2.txt)
for f in old_1 new_1 old_2 new_2; do echo "$f: $(tail -n 1 gpurun_out/r5ab/$f.txt)"; done
