mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
