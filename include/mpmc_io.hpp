// include/mpmc_io.hpp -- the reference's two on-disk formats for the C++ facade, hot-path subset.
//
//   read_pqr    PQR geometry, token grammar of reference src/System.cpp:583-700:
//               ATOM id type moltype FLAG molid x y z mass charge[e] alpha eps sigma omega gwp_alpha [c6 c8 c10 c9]
//               rows with molecule type BOX are skipped (:592), END stops (:590), charge *= 408.7816 (:624),
//               a new molecule starts when molid changes (:672), FLAG F = frozen (:599-606).
//   read_input  "keyword value..." input file (src/SimulationControl.cpp:204-267), case-insensitive, '!' / '#' comments.
//               Hot-path keywords set the facade's fields; keywords that switch on physics outside the energy hot path
//               set the matching MPMC_FLAG_* bit (the library then refuses with unsupported_setting = 4004).
//   write_pqr   rows in the layout of the reference's writer (src/System.Output.cpp:731-768, restart precision).
// Errors are thrown as int with the reference's codes (fopen_fail_read 1000, invalid_input 3000, invalid_datum 6001 ...).
#pragma once

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "mpmc_system.hpp"

namespace mpmc {

constexpr double E2REDUCED = 408.7816; // reference src/constants.h:35

namespace io_detail {
inline std::string lower(std::string s) {
	std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
	return s;
}
inline std::vector<std::string> tokens(const std::string &line) {
	std::istringstream is(line);
	std::vector<std::string> t;
	std::string w;
	while (is >> w) t.push_back(w);
	return t;
}
inline bool to_double(const std::string &s, double &v) {
	char *end = nullptr;
	v = std::strtod(s.c_str(), &end);
	return end != s.c_str() && *end == '\0';
}
inline int onoff(const std::string &v) {
	const std::string l = lower(v);
	if (l == "on") return 1;
	if (l == "off") return 0;
	throw 3000; // invalid_input
}
inline std::string dirname_of(const std::string &p) {
	const size_t k = p.find_last_of('/');
	return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
} // namespace io_detail

inline void read_pqr(const std::string &path, System &s) {
	using namespace io_detail;
	std::ifstream f(path);
	if (!f) throw 1000; // fopen_fail_read
	s.atoms.clear();
	s.moltype_names.clear();
	std::string line;
	bool have_mol = false;
	long cur_molid = 0;
	int mol_index = -1;
	while (std::getline(f, line)) {
		const std::vector<std::string> t = tokens(line);
		if (t.empty()) continue;
		if (lower(t[0]).compare(0, 3, "end") == 0) break;
		if (lower(t[0]) != "atom") continue;
		if (t.size() > 3 && lower(t[3]) == "box") continue; // box-corner pseudo atoms carry a short row
		if (t.size() < 16) throw 6000; // missing_required_datum
		double v[11];
		const int idx[11] = {6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 5};
		for (int k = 0; k < 11; k++)
			if (!to_double(t[idx[k]], v[k])) throw 6001; // invalid_datum
		const long molid = (long)v[10];
		if (!have_mol || molid != cur_molid) {
			have_mol = true;
			cur_molid = molid;
			mol_index++;
		}
		const std::string flag = lower(t[4]);
		if (flag == "a" || flag == "s" || flag == "t") throw 4004; // adiabatic / spectre / target: outside the hot path
		Atom a;
		a.pos[0] = v[0];
		a.pos[1] = v[1];
		a.pos[2] = v[2];
		a.mass = v[3];
		a.charge = v[4] * E2REDUCED;
		a.polarizability = v[5];
		a.epsilon = v[6];
		a.sigma = v[7];
		a.frozen = (flag == "f") ? 1 : 0;
		a.molecule = mol_index;
		for (size_t k = 0; k < s.moltype_names.size() && a.moltype < 0; k++)
			if (s.moltype_names[k] == t[3]) a.moltype = (int)k;
		if (a.moltype < 0) {
			s.moltype_names.push_back(t[3]);
			a.moltype = (int)s.moltype_names.size() - 1;
		}
		double c;
		if (t.size() > 16 && to_double(t[16], c)) a.c6 = c;
		if (t.size() > 17 && to_double(t[17], c)) a.c8 = c;
		if (t.size() > 18 && to_double(t[18], c)) a.c10 = c;
		s.atoms.push_back(a);
	}
	if (s.atoms.empty()) throw 3001; // no_molecules_in_system
	s.atoms_changed();
}

// returns the path of the PQR file named by pqr_input (relative names are resolved next to the input file)
inline std::string read_input(const std::string &path, System &s) {
	using namespace io_detail;
	std::ifstream f(path);
	if (!f) throw 1000;
	static const std::map<std::string, uint64_t> unsupported = {
	    {"rd_crystal", MPMC_FLAG_RD_CRYSTAL}, {"spectre", MPMC_FLAG_SPECTRE},
	    {"gwp", MPMC_FLAG_GWP}, {"sg", MPMC_FLAG_USE_SG}, {"polarvdw", MPMC_FLAG_POLARVDW}, {"cdvdw", MPMC_FLAG_POLARVDW},
	    {"polar_ewald_full", MPMC_FLAG_POLAR_EWALD_FULL}, {"polar_wolf", MPMC_FLAG_POLAR_WOLF}, {"polar_wolf_full", MPMC_FLAG_POLAR_WOLF},
	    {"polar_palmo", MPMC_FLAG_POLAR_PALMO}, {"polar_gs_ranked", MPMC_FLAG_POLAR_GS_RANKED}, {"polar_sor", MPMC_FLAG_POLAR_SOR},
	    {"polar_esor", MPMC_FLAG_POLAR_SOR}, {"polar_zodid", MPMC_FLAG_POLAR_ZODID}, {"waldmanhagler", MPMC_FLAG_NON_LB_MIXING},
	    {"halgren_mixing", MPMC_FLAG_NON_LB_MIXING}, {"c6_mixing", MPMC_FLAG_NON_LB_MIXING}, {"dreiding", MPMC_FLAG_OTHER_RD},
	    {"lj_buffered_14_7", MPMC_FLAG_OTHER_RD}, {"disp_expansion", MPMC_FLAG_OTHER_RD}, {"rd_anharmonic", MPMC_FLAG_OTHER_RD},
	    {"axilrod_teller", MPMC_FLAG_AXILROD_TELLER}, {"cavity_autoreject", MPMC_FLAG_CAVITY_AUTOREJECT},
	    {"cavity_autoreject_absolute", MPMC_FLAG_CAVITY_AUTOREJECT}};
	std::string pqr, line;
	while (std::getline(f, line)) {
		const std::vector<std::string> t = tokens(line);
		if (t.empty() || t[0][0] == '!' || t[0][0] == '#') continue;
		const std::string k = lower(t[0]);
		auto need = [&](size_t n) {
			if (t.size() < n + 1) throw 3000;
		};
		auto dval = [&](size_t i) {
			double v;
			if (!to_double(t[i], v)) throw 3000;
			return v;
		};
		if (k == "basis1" || k == "basis2" || k == "basis3") {
			need(3);
			const int r = k[5] - '1';
			for (int c = 0; c < 3; c++) s.pbc.basis[r][c] = dval(1 + c);
		} else if (k == "pqr_input") {
			need(1);
			pqr = t[1];
		} else if (k == "rd_only") { need(1); s.rd_only = onoff(t[1]); }
		else if (k == "rd_lrc") { need(1); s.rd_lrc = onoff(t[1]); }
		else if (k == "polarization") { need(1); s.polarization = onoff(t[1]); }
		else if (k == "polar_iterative") { need(1); s.polar_iterative = onoff(t[1]); }
		else if (k == "polar_ewald") { need(1); s.polar_ewald = onoff(t[1]); }
		else if (k == "polar_gs") { need(1); s.polar_gs = onoff(t[1]); }
		else if (k == "polar_rrms") { need(1); s.polar_rrms = onoff(t[1]); }
		else if (k == "wolf") { need(1); s.wolf = onoff(t[1]); }
		else if (k == "feynman_hibbs") { need(1); s.feynman_hibbs = onoff(t[1]); }
		else if (k == "feynman_hibbs_order") { need(1); s.feynman_hibbs_order = (int)dval(1); }
		else if (k == "temperature") { need(1); s.temperature = dval(1); }
		else if (k == "polar_max_iter") { need(1); s.polar_max_iter = (int)dval(1); }
		else if (k == "ewald_kmax") { need(1); s.ewald_kmax = (int)dval(1); }
		else if (k == "polar_precision") { need(1); s.polar_precision = dval(1); }
		else if (k == "polar_gamma") { need(1); s.polar_gamma = dval(1); }
		else if (k == "polar_damp") { need(1); s.polar_damp = dval(1); }
		else if (k == "ewald_alpha") { need(1); s.ewald_alpha = dval(1); s.ewald_alpha_set = 1; }
		else if (k == "polar_ewald_alpha") { need(1); s.polar_ewald_alpha = dval(1); s.polar_ewald_alpha_set = 1; }
		else if (k == "polar_damp_type") {
			need(1);
			const std::string v = lower(t[1]);
			s.damp_type = (v == "exponential") ? DAMPING_EXPONENTIAL : (v == "linear") ? DAMPING_LINEAR : DAMPING_OFF;
		} else {
			auto it = unsupported.find(k);
			if (it != unsupported.end() && (t.size() < 2 || lower(t[1]) != "off")) s.unsupported_flags |= it->second;
			// everything else (job_name, ensemble, temperature, numsteps, output switches ...) does not enter energy()
		}
	}
	if (pqr.empty()) throw 4003; // missing_setting
	if (pqr[0] != '/') pqr = dirname_of(path) + "/" + pqr;
	return pqr;
}

inline void load_system(const std::string &input_path, System &s) {
	const std::string pqr = read_input(input_path, s);
	read_pqr(pqr, s);
	s.update_pbc();
}

inline void write_pqr(const std::string &path, const System &s) {
	FILE *fp = std::fopen(path.c_str(), "w");
	if (!fp) throw 1001; // fopen_fail_write
	for (size_t i = 0; i < s.atoms.size(); i++) {
		const Atom &a = s.atoms[i];
		std::fprintf(fp, "ATOM  %5d %-4.45s %-3.3s %-1.1s%4d    %11.6f %11.6f %11.6f  %8.4f %8.4f %8.5f %8.5f %8.5f %8.5f %8.5f %8.5f %8.5f %8.5f\n",
		             (int)i + 1, "X", "M", a.frozen ? "F" : "M", a.molecule + 1, a.pos[0], a.pos[1], a.pos[2], a.mass, a.charge / E2REDUCED,
		             a.polarizability, a.epsilon, a.sigma, 0.0, 0.0, a.c6, a.c8, a.c10);
	}
	std::fprintf(fp, "END\n");
	std::fclose(fp);
}

} // namespace mpmc
