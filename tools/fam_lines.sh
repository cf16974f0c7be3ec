o=gpurun_out/fam2; mkdir -p $o
for f in diag-l1box-box diag-nonneg-box diag-indbox-box diag-indboxvec-box diag-l1-boxvec diag-zero-boxveclo diag-l1-free diag-l1-zero diag-zero-vc diag-l1-cc diag-nonneg-eitheror diag-l1-xor; do
  python bench.py --family $f --no-extras --no-cpu-baseline > $o/bench_family_$f.json 2> /dev/null; echo "$f rc=$?"
done
python bench.py --no-extras --no-cpu-baseline > $o/bench_cfg2.json 2>/dev/null
python tools/bench_print.py $o/*.json
