#!/bin/bash
# same-box A/B of prebuilt library variants (box-to-box variance is +-3 %): tools/ab_libs.sh tools/probe/lib_A.so tools/probe/lib_B.so ...
# runs tools/quick_merge.sh (kernel trace of the C3 and C2 shapes) once per variant, twice round-robin
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  for v in "$@"; do
    export TK_HIP_LIB=$R/$v   # (the shipped library is never overwritten: tekken-rs_amd/__init__.py loads what TK_HIP_LIB names)
    echo "== $v (rep $rep)"
    bash tools/quick_merge.sh | grep "^c2" || exit 1
  done
done
