for sp in 3 4 5 6 8 10 12; do echo "== splits $sp"; PRMF_SWEEP_SPLITS=$sp python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400; done
