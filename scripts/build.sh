#!/bin/bash
# Builds libjuqbox_hip.so and the oracle and REPORTS the outcome (exit code, error / warning lines): `make | grep` hides a failed
# build behind an empty output, and a stale .so then travels to the GPU box.
cd "$(dirname "$0")/../juqbox.jl_amd/csrc" || exit 1
make -j8 > /tmp/jq_make.log 2>&1
rc=$?
grep -E "error|warning:|rebuilt without" /tmp/jq_make.log | grep -v "^[[:space:]]*{ echo" | cut -c1-300
make -C ../../oracle >> /tmp/jq_make.log 2>&1 || rc=1
echo "build rc=$rc  $(python3 - <<'PY'
import ctypes, os
L = ctypes.CDLL(os.path.join(os.getcwd(), "..", "libjuqbox_hip.so"))
L.jq_version.restype = ctypes.c_char_p
print(L.jq_version().decode())
PY
)"
exit $rc
