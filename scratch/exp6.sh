python -m pytest tests/test_gpu_parity.py tests/test_gpu_medium.py -m gpu -x -q 2>&1 | tail -3
python scratch/race_probe.py 2>&1 | grep -E "differs|FPE" | head -4
for cfg in "8 1 2" "8 1 3" "8 1 4" "8 0 3" "8 0 4" "4 1 3" "4 1 6" "4 0 5" "4 0 8"; do set -- $cfg
  export PRMF_SWEEP_NW=$1 PRMF_SWEEP_BDB=$2 PRMF_SWEEP_WGS_PER_CU=$3
  echo -n "NW=$1 BDB=$2 WGS/CU=$3: "; python scratch/time_sweep.py 2>&1 | tail -1
done
