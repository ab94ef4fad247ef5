#!/bin/bash
echo "=== old plan"; SRCNN_DEBUG_PLAN=1 python tools/diag_light.py 2>&1 | grep -v amdgpu.ids
echo "=== balanced"; python tools/diag_light.py 2>&1 | grep -v amdgpu.ids
