# same-box A/B of several builds of the kernel library on the contraction shapes (tools/w256_bench.py, two alternating rounds)
#   bash tools/r4_ab_multi.sh TAG name1 name2 ...      (names of interactron_amd/lib/libab_<name>.so; "intree" = the product library)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for i in 1 2; do
  for n in "$@"; do
    if [ "$n" = intree ]; then python tools/w256_bench.py > gpurun_out/$TAG/${n}_$i.txt 2>&1
    else IX_LIB_PATH=$PWD/interactron_amd/lib/libab_$n.so python tools/w256_bench.py > gpurun_out/$TAG/${n}_$i.txt 2>&1; fi
  done
done
for n in "$@"; do for i in 1 2; do echo "$n run $i: $(tail -n 1 gpurun_out/$TAG/${n}_$i.txt)"; done; done
python3 - "$TAG" "$@" <<'PY'
import sys
tag, names = sys.argv[1], sys.argv[2:]
rows = {}
for n in names:
    for i in (1, 2):
        for line in open("gpurun_out/%s/%s_%d.txt" % (tag, n, i)):
            f = line.split()
            if len(f) >= 9 and f[0].isdigit():
                rows.setdefault(" ".join(f[:6]), {}).setdefault(n, []).append(float(f[6]))
print("%-30s" % "128 x 128 tiles, best of runs, us" + " ".join("%10s" % n for n in names))
for k, v in rows.items():
    print("%-30s" % k + " ".join("%10.1f" % min(v.get(n, [0])) for n in names))
PY
