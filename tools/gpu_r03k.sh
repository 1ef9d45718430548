mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -20
