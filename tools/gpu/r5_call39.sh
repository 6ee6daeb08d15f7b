#!/bin/bash
# chain parity (front end + solver against the oracle chain) on the final library, under a hard limit
cd /root/repo
(timeout -s KILL 55 python -m pytest tests/test_chain_parity.py tests/test_dropin_sequence.py -x -q -m gpu -k "chain_parity_gpu or product_backend_gpu" 2>&1 | tail -3) > gpurun_out/r5_final_chain_parity.txt
