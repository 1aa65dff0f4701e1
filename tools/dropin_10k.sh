#!/bin/bash
# the reference's own PI driver on the 10 000-atom polarizable BASELINE box (P = 4, a few steps) with MPMC_WRAP_MODE=both: every
# energy() call is evaluated by the reference too (~9 s and ~17 GB per image on the CPU) and compared component by component at 1e-9
# usage (GPU box): bash tools/dropin_10k.sh [steps]
root=${GRAFT_REPO_ROOT:-$(pwd)}
steps=${1:-2}
d=$(mktemp -d)
python3 - "$d" <<PY
import sys
sys.path.insert(0, "$root")
from mpmcxx_amd import gen_box
inp, pqr = gen_box.materialize("ion10k_polar", sys.argv[1])
txt = open(inp).read().splitlines()
keep = [l for l in txt if not l.split()[0] in ("ensemble", "numsteps", "corrtime", "seed", "move_factor", "rot_factor", "job_name", "temperature", "pqr_restart", "pqr_output", "energy_output", "dipole_output", "field_output")]
open(sys.argv[1] + "/pi.in", "w").write("job_name big\nensemble pi_nvt\ntemperature 80.0\nnumsteps $steps\ncorrtime 1\nseed 3\nmove_factor 0.01\nrot_factor 1.0\n"
    "bead_perturb_probability 0.5\nPI_trial_chain_length 2\nwrapall on\nparallel_restarts off\n" + "\n".join(keep) + "\n")
PY
cd $d
s=$(date +%s)
MPMC_WRAP_MODE=both OMP_NUM_THREADS=4 timeout 1000 $root/oracle/_ref/mpmcxx_wrapped -P 4 pi.in > out.log 2> err.log; rc=$?
echo "rc=$rc after $(( $(date +%s) - s )) s"; grep -h "ref_adapter" err.log | cut -c1-300; grep -v "^#" big.energy.dat | cut -c1-110
