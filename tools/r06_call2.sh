R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
python tools/tradeoff_digits.py > $OUT/tradeoff_digits.txt 2>&1; echo "digits rc=$?"; cat $OUT/tradeoff_digits.txt | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_nnls_matrix.py tests/test_gpu_textio.py -m gpu -x -q > $OUT/t3.txt 2>&1; echo "tests rc=$?"; tail -5 $OUT/t3.txt
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06/bench_default.json")).read().strip().splitlines()[-1])
r=d["roofline"]
print("C3 %.1f it/s %.4f ms | bound %s frac %.3f | hbm %.3f mfma %.3f (clock %s) | stream %.0f of-stream %.3f | traffic %s" % (d["value"], d["ms_per_step"], r["bound"], r["frac"], r["hbm"]["frac"], r["mfma"]["frac"], r["mfma"]["clock_ghz"], r.get("stream_gbs",0), r.get("frac_of_stream",0), r["traffic"]))
o=d["outer_loop"]; print("outer: first %.3f warm %.3f (%.0f it/s) dev-resident %.3f (%.0f it/s) inner %.3f" % (o["seconds_first_call"], o["seconds_warm_call"], o["it_per_s_warm"], o["seconds_device_resident"], o["it_per_s_device_resident"], o["seconds_inner_iterations"]))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["lean_value"], d["cpu_baseline"]["cores"])
for k,v in d["secondary"].items(): print(k, json.dumps({a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a not in ("note",)}))
PY
