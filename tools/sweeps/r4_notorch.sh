ulimit -c 0
for w in human yeast ecoli adversarial; do
  BARGS="--no-side-legs" WLS="$w" HWQS="16" bash tools/r4_bench_x.sh
  echo "no torch:"; BARGS="--no-side-legs --no-torch" WLS="$w" HWQS="16" bash tools/r4_bench_x.sh
done
