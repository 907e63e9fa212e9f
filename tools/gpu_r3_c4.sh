set -x
O=gpurun_out/c4
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_host_api.py tests/test_flagship_golden.py tests/test_train_traces.py tests/test_gpi_agent.py tests/test_shape_fuzz.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 > $O/tests.log
python tools/chain_latency.py > $O/latency_chain4.txt 2>&1
MORL_CHAIN4=0 python tools/chain_latency.py > $O/latency_chain16.txt 2>&1
python tools/diag_single_pass.py 1355 >> $O/latency_chain4.txt 2>&1
MORL_CHAIN4=0 python tools/diag_single_pass.py 1355 >> $O/latency_chain16.txt 2>&1
python tools/diag_single_pass.py 2048 >> $O/latency_chain4.txt 2>&1
MORL_CHAIN4=0 python tools/diag_single_pass.py 2048 >> $O/latency_chain16.txt 2>&1
python tools/diag_single_pass.py 8 >> $O/latency_chain4.txt 2>&1
MORL_CHAIN4=0 python tools/diag_single_pass.py 8 >> $O/latency_chain16.txt 2>&1
