#!/bin/bash
# call 39: ADA forward / adjoint time against the reflect margin (identity maps)
cd "$GRAFT_REPO_ROOT"
python tools/ada_margin_probe.py 2>&1 | tail -12
