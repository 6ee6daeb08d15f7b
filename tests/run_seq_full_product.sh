set -x
export PVIO_CHAIN_REPORT=$PWD/gpurun_out/r4_chain_parity.json
python -m pytest tests/test_gpu_klt.py tests/test_chain_parity.py tests/test_golden.py tests/test_host_frontend.py -m gpu -x -q -s 2>&1 | grep -v "^chain ok" | tail -25 > gpurun_out/r4_klt_chain_gpu.txt
# the reference's pvio::PVIO with the FULL product (HipImage + product back-end) against the same with the reference's back-end + oracle front end
mkdir -p /tmp/s && PVIO_SEQ_IMAGE=oracle python tests/chain_run.py oracle/_ref/libpvio_ref.so /tmp/s/ref 60 6 3 25.0 full | tail -1
PVIO_SEQ_IMAGE=hip python tests/chain_run.py oracle/_ref/libpvio_dropin.so /tmp/s/hip 60 6 3 25.0 full | tail -1
python - <<'PY' > gpurun_out/r4_seq_full_product.txt 2>&1
import sys; sys.path.insert(0,'tests')
import chain_compare, test_host_headless as hh, json
info = chain_compare.compare_seq('/tmp/s/ref.log','/tmp/s/hip.log', hh.K4[0])
print(json.dumps(info, indent=1, default=float))
PY
tail -5 gpurun_out/r4_seq_full_product.txt
