#!/bin/bash
# usage: tools/gpu_ab.sh <ab-name> <python script + args>: the script with the default library, then with the A/B library
cd "${GRAFT_REPO_ROOT:-.}"
ab=$1; shift
echo "== default"; timeout -k 10 500 python "$@" 2>&1 | grep -v balance
echo "== $ab"; PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$ab.so timeout -k 10 500 python "$@" 2>&1 | grep -v balance
