#!/bin/bash
# In-kernel cycle stamps of the Winograd form (GPU box): builds a trace-enabled copy of the library into a scratch dir, runs tools/trace_conv.py on it.
set -e
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/trace_build && mkdir -p /tmp/trace_build && cp -r maxstyle_amd include tools /tmp/trace_build/
cd /tmp/trace_build/maxstyle_amd/csrc && rm -rf build && touch ms_conv_wide.h && make -j16 EXTRA=-DMS_CONV_TRACE_BUILD > /tmp/trace_build/make.log 2>&1 || { tail -20 /tmp/trace_build/make.log; exit 1; }
cd /tmp/trace_build
if [ -n "$1" ]; then shift 0; python tools/trace_conv.py "$@" 2>&1 | grep -v amdgpu; exit 0; fi
for m in plain bwd; do echo "== $m (Winograd form)"; MS_OPTIONS=conv.wino=2 python tools/trace_conv.py $m 2>&1 | grep -v amdgpu; done
echo "== plain (direct form)"; MS_OPTIONS=conv.wino=0 python tools/trace_conv.py plain 2>&1 | grep -v amdgpu | head -16
