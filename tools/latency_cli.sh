#!/bin/bash
# C-ABI latency of one energy() call, measured from C++ (examples/energy_cli --time): no Python in the loop.
# usage (GPU box, repo root): bash tools/latency_cli.sh [reps]
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
reps=${1:-2000}
g++ -std=c++14 -O2 -I $root/include $root/examples/energy_cli.cpp -L $root/mpmcxx_amd -lmpmc_energy -Wl,-rpath,$root/mpmcxx_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/energy_cli
for f in lj64 lj1000 ion64_es ion216_polar ion216_precision ion1000_polar; do
	echo -n "$f default:            "; /tmp/energy_cli $root/tests/golden/$f.in --time $reps | tail -1
done
# (the general multi-kernel path of small LJ boxes: python -c "energy.configure('single_launch', 0)" through tools/latency_probe.py)
