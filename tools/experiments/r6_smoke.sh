#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
