#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 2400 python tools/routing_sweep.py > gpurun_out/r05_routing_sweep.json 2> gpurun_out/r05_routing_sweep.err; echo rc=$?; tail -c 1500 gpurun_out/r05_routing_sweep.json
