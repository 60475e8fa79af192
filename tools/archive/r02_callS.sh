OUT=gpurun_out/r02s; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -q -rA -p no:cacheprovider -x > $OUT/pytest.log 2>&1; rc=$?; echo pytest_rc=$rc; tail -3 $OUT/pytest.log | cut -c1-200
if [ $rc -ne 0 ]; then echo "tests failed: no further GPU step"; exit 1; fi
sed -i 's#OUT=gpurun_out/r02r#OUT=gpurun_out/r02s#' tools/r02_callR.sh
bash tools/r02_callR.sh
