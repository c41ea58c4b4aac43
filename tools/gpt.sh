#!/bin/bash
# run pytest on the GPU box with the library-banner noise filtered out: tools/gpt.sh <pytest args>
timeout 900 python -m pytest "$@" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path\|amdgpu.ids" | tail -15
