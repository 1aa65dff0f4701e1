#!/bin/bash
# oracle/make_pi_golden.sh -- regenerates the stock-binary path-integral goldens under tests/golden/{pi001,pi_ion27,pi_h2,pi_water64,pi_frozen,pi_tri,pi_nopbc,pi_wolf,pi_gs,pi_ion1000,pi_h2_orient}
# by running the UNMODIFIED reference executable (oracle/_ref/mpmcxx, built in place by `make -C oracle ref`) on the
# committed inputs.  Test infrastructure; needs /root/reference (container only).  Outputs are data: the energy / dipole /
# field traces, the final averages block and the final bead geometries ("long_output on" = %.6f coordinates).
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
G=$here/../tests/golden
BIN=$here/_ref/mpmcxx
run_case() { # dir input P job
	local d
	d=$(mktemp -d)
	cp "$G/$1"/*.in "$G/$1"/*.pqr "$d"/
	rm -f "$d"/golden_*
	(cd "$d" && "$BIN" -P "$3" "$2" >stock.log 2>&1)
	grep -v '^#' "$d/$4.energy.dat" >/dev/null
	cp "$d/$4.energy.dat" "$G/$1/golden_energy.dat"
	if [ "${5:-finals}" = finals ]; then for f in "$d/$4".final-*.pqr; do cp "$f" "$G/$1/golden_${f##*/$4.}"; done; fi
	if [ "${5:-finals}" = finals ]; then for k in dipole field; do [ -s "$d/$4.$k.dat" ] && cp "$d/$4.$k.dat" "$G/$1/golden_$k.dat"; done; fi
	grep -E '^OUTPUT: (AR =|total energy|kinetic energy|polarization energy)' "$d/stock.log" | tail -12 >"$G/$1/golden_final_averages.txt"
	echo "$1: $(grep -vc '^#' "$G/$1/golden_energy.dat") energy rows"
	rm -rf "$d"
}
run_case pi001 equilibrate.in 8 ArAr2K
run_case pi_ion27 input.in 4 ion27
run_case pi_h2 input.in 4 h2pi   # 8 rigid diatomics: the rotation of PI_displace and the rigid translation of the bead moves
run_case pi_water64 input.in 4 water64   # 64 rigid 3-site polarizable molecules + one neutral atom: exclusions, rotation, Ewald, Thole together
run_case pi_frozen input.in 4 frozen   # a frozen charged framework (27 sites, flag F) + 6 mobile polar diatomics: frozen pairs, moves pick movable molecules only
run_case pi_tri input.in 4 tri        # triclinic cell, Jacobi terminated by polar_precision
run_case pi_nopbc input.in 4 nopbc    # static field without Ewald (thole_field_nopbc), polar_gamma, dipole rrms
run_case pi_wolf input.in 4 wolf      # Wolf electrostatics, no LRC
run_case pi_gs input.in 4 gs          # Gauss-Seidel sweeps + dipole rrms
run_case pi_h2_orient input.in 4 h2or   # orientational bead moves: per-image restart files with scattered orientations (oracle/make_pi_orient_fixture.py), sorbate_* keys
run_case pi_ion1000 input.in 4 ion1000 rows-only   # 1000 polarizable ions: energy.dat rows and averages only (size)
