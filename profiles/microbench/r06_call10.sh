python -m pytest tests/test_depgraph_dev.py tests/test_jni_shim.py -q -x 2>&1 | tail -4
python -m pytest tests/test_epaxos.py tests/test_gpu_fullsize.py -q -x -k "epaxos or execute or depgraph" 2>&1 | tail -3
