#!/bin/bash
# The GPU tests under guard pages (tests/guard_pages.py): one pytest process per file - a GPU memory fault aborts the process, the next file still runs.
#   bash tools/guard_run.sh end|start [files...]        (on the GPU box; prints one line per file)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
MODE=${1:-end}; shift
FILES=${*:-"tests/test_wino_gpu.py tests/test_conv_gpu.py tests/test_k3n_gpu.py tests/test_k1s_gpu.py tests/test_k9_gpu.py tests/test_s2pro_gpu.py tests/test_gen2_random_gpu.py tests/test_actbwd_xfin_gpu.py tests/test_wgrad_gpu.py tests/test_maxstyle_gpu.py tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_round6_gpu.py"}
for f in $FILES; do
  log=$(mktemp)
  MS_GUARD_PAGES=$MODE PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python -m pytest $f -q -m gpu -x -k "not teacher_forced" > "$log" 2>&1; rc=$?
  if [ $rc -eq 0 ]; then echo "guard=$MODE $f: $(grep -E 'passed|failed' "$log" | tail -1)"; else echo "guard=$MODE $f: EXIT CODE $rc"; grep -E "Memory access fault|^FAILED|Error|error:|test_.*\.py\", line" "$log" | head -8 | cut -c1-240; fi
  rm -f "$log"
done
