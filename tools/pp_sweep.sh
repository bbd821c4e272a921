python -m pytest tests/test_ops_gpu.py -q -x -k "gate_residual or fused_adaln_chain" 2>&1 | tail -3 && ONLY=OUT,FF2 python tools/gemm_tune.py 938 33,43,35,45
