python -m pytest tests/test_optimizer.py tests/test_model_parity.py -m gpu -x -q -k "checkpoint or egomcq" 2>&1 | grep -v Warn | tail -8
