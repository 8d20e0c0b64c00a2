#!/bin/bash
# usage: tools/gpu_ab_multi.sh "<ab names>" <python script + args>: the script with the default library, then with each A/B library
cd "${GRAFT_REPO_ROOT:-.}"
abs=$1; shift
echo "== default"; timeout -k 10 400 python "$@" 2>&1 | grep -v balance
for ab in $abs; do
  echo "== $ab"; PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$ab.so timeout -k 10 400 python "$@" 2>&1 | grep -v balance || exit 1
done
