#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for lib in "$@"; do echo "== $lib"; D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib python tools/stage_times.py 2>/dev/null | grep -E "^(ALL |RAYCAST|DYNGRID|TRACKER)" | head -4; done
