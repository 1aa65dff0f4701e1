"""What the reference's OTHER callers of System::energy() ask of the path: `System::mc` in the uVT / NPT ensembles and `Gibbs_mc`
(src/System.MonteCarlo.cpp:21, src/SimulationControl.Gibbs.cpp:133) change the number of molecules (insert / remove,
src/System.MonteCarlo.cpp:952-1104) and the cell (volume_change, :1287-1338: basis scaled by (V'/V)^(1/3), every molecule shifted
by the scaled centre of mass) between two evaluations of the same System.

Those loops cannot be run from the reference in this image (its non-MPI build aborts them: `size` stays 0 in setup_mpi_dataStructs,
src/System.MonteCarlo.cpp:211-245, and do_corrtime_bookkeeping writes through a null `mpi_data.temperature`, :1978; the Gibbs loop
prints NaN averages and crashes), so this drives the same sequence of state changes through ONE live context per box -- two boxes, a
coupled volume move, a particle transfer, displacements, rejections that restore the previous state -- and checks every evaluation
against the oracle evaluated from scratch on the same state."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy
from oracle import OracleSystem

pytestmark = pytest.mark.gpu

KEYS = ("rd_energy", "coulombic_energy", "polarization_energy", "energy")


class Box:
    """host-side state of one simulation box + its live device context"""

    def __init__(self, atoms, basis, opts):
        self.atoms = {k: np.array(v) for k, v in atoms.items()}
        self.basis = np.array(basis, dtype=np.float64)
        self.opts = opts
        self.sys = energy.System(self.atoms, self.basis, opts)  # capacity = the initial atom count: insertions make the context grow

    def snapshot(self):
        return {k: v.copy() for k, v in self.atoms.items()}, self.basis.copy()

    def restore(self, snap):
        self.atoms, self.basis = {k: v.copy() for k, v in snap[0].items()}, snap[1].copy()
        self.sys.set_box(self.basis)
        self.sys.set_atoms(self.atoms)

    def n(self):
        return len(self.atoms["charge"])

    def molecules(self):
        return np.unique(self.atoms["mol_id"])

    def scale_volume(self, factor):
        """volume_change: basis *= s, every molecule translated by (s - 1) * com"""
        s = factor ** (1.0 / 3.0)
        self.basis = self.basis * s
        pos, mol, mass = self.atoms["pos"], self.atoms["mol_id"], self.atoms["mass"]
        for m in self.molecules():
            sel = mol == m
            com = (pos[sel] * mass[sel, None]).sum(0) / mass[sel].sum()
            pos[sel] += com * s - com
        self.sys.set_box(self.basis)
        self.sys.update_positions(0, pos)

    def remove_molecule(self, m):
        keep = self.atoms["mol_id"] != m
        gone = {k: v[~keep].copy() for k, v in self.atoms.items()}
        self.atoms = {k: v[keep].copy() for k, v in self.atoms.items()}
        # enumerate_particles: molecule ids stay dense
        _, self.atoms["mol_id"] = np.unique(self.atoms["mol_id"], return_inverse=True)
        self.atoms["mol_id"] = self.atoms["mol_id"].astype(np.int32)
        self.sys.set_atoms(self.atoms)
        return gone

    def insert_molecule(self, mol_atoms, com_new, at_index):
        """insert a copy of `mol_atoms` with its centre of mass at com_new, in front of atom `at_index` (list insertion, :1051-1062)"""
        add = {k: v.copy() for k, v in mol_atoms.items()}
        com = (add["pos"] * add["mass"][:, None]).sum(0) / add["mass"].sum()
        add["pos"] = add["pos"] + (com_new - com)
        add["mol_id"] = np.full(len(add["charge"]), -1, dtype=np.int32)
        merged = {k: np.concatenate([self.atoms[k][:at_index], add[k], self.atoms[k][at_index:]]) for k in self.atoms}
        # renumber molecules in list order
        ids, first = [], {}
        for i, m in enumerate(merged["mol_id"]):
            key = ("new",) if m == -1 else ("old", int(m))
            first.setdefault(key, len(first))
            ids.append(first[key])
        merged["mol_id"] = np.array(ids, dtype=np.int32)
        self.atoms = merged
        self.sys.set_atoms(self.atoms)

    def displace(self, m, delta):
        sel = np.nonzero(self.atoms["mol_id"] == m)[0]
        self.atoms["pos"][sel] += delta
        self.sys.update_positions(int(sel[0]), self.atoms["pos"][sel])  # molecules are contiguous

    def check(self, label):
        self.sys.energy()
        got = self.sys.observables
        ref = OracleSystem(self.atoms, self.basis, self.opts).energy()
        for k in KEYS:
            assert abs(got[k] - ref[k]) <= 1e-9 * max(abs(ref[k]), 1e-3 * abs(ref["energy"])), (label, k, got[k], ref[k])
        assert int(got["n_lj_in_cutoff"]) == int(ref["n_lj_in_cutoff"]) and int(got["n_es_in_cutoff"]) == int(ref["n_es_in_cutoff"]), label
        if self.opts.get("polarization"):
            mu = self.sys.dipoles()[0]
            assert mu.shape == ref["mu"].shape
            assert np.abs(mu - ref["mu"]).max() <= 1e-9 * np.abs(ref["mu"]).max() + 1e-13, label
        return got["energy"]

    def close(self):
        self.sys.close()


def two_boxes(name):
    """box A = the fixture, box B = the same cell holding the first half of its molecules (the vapour side)"""
    a, basis, opts = util.load_fixture(name)
    mols = np.unique(a["mol_id"])
    keep = np.isin(a["mol_id"], mols[: len(mols) // 2 // 2 * 2])  # an even number of ions keeps box B neutral
    b = {k: v[keep].copy() for k, v in a.items()}
    return Box(a, basis, opts), Box(b, basis.copy(), opts)


@pytest.mark.parametrize("fixture", ["ion216_polar", "ion64_es", "lj1000", "water64_polar", "ion1000_polar"])
def test_gibbs_style_move_sequence_on_two_live_contexts(fixture):
    rng = np.random.default_rng(11)
    A, B = two_boxes(fixture)
    boxes = (A, B)
    for bx in boxes:
        bx.check("initial")
    for step in range(12):
        kind = ("volume", "transfer", "displace")[step % 3]
        snaps = [bx.snapshot() for bx in boxes]
        if kind == "volume":
            # coupled: V_A' = V_A * f, V_B' = V_B + V_A - V_A'  (volume_change_Gibbs :1296-1304)
            va, vb = abs(np.linalg.det(A.basis)), abs(np.linalg.det(B.basis))
            f = np.exp((rng.random() - 0.5) * 0.1)
            A.scale_volume(f)
            B.scale_volume((vb + va - va * f) / vb)
        elif kind == "transfer":
            src, dst = (A, B) if rng.random() < 0.5 else (B, A)
            m = int(rng.choice(src.molecules()))
            gone = src.remove_molecule(m)
            while True:  # a random point of the cell (:1009-1016) that is not a bad contact (those are rejected before they matter)
                frac = 0.5 - rng.random(3)
                d = (dst.atoms["pos"] - frac @ dst.basis) @ np.linalg.inv(dst.basis)
                if np.linalg.norm((d - np.rint(d)) @ dst.basis, axis=1).min() > 2.5:
                    break
            dst.insert_molecule(gone, frac @ dst.basis, at_index=int(rng.choice(np.nonzero(np.diff(dst.atoms["mol_id"], prepend=-1))[0])))
        else:
            for bx in boxes:
                bx.displace(int(rng.choice(bx.molecules())), rng.normal(scale=0.2, size=3))
        for i, bx in enumerate(boxes):
            bx.check(f"step {step} {kind} box {i}")
        if step % 2:  # reject: restore() puts the previous molecules and cell back
            for bx, s in zip(boxes, snaps):
                bx.restore(s)
                bx.check(f"step {step} {kind} restored")
    n_total = A.n() + B.n()
    assert n_total == sum(len(s[0]["charge"]) for s in snaps)  # transfers conserve atoms
    for bx in boxes:
        bx.close()


@pytest.mark.parametrize("name", ["water64_polar", "ion1000_polar"])
def test_order_carried_across_insertions_and_removals(name):
    """The spatial order is carried across a contiguous insertion / removal (context.cpp carry_spatial_order) instead of re-sorting per
    move; every 64 atom edits a real sort follows.  Any permutation is the same physics: after each of 45 random insert / remove moves
    at random places of the molecule list (System::insert / remove semantics, src/System.MonteCarlo.cpp:1051-1062) the live context must
    agree with a FRESH context on the same list -- energies to 1e-11, every pair count exactly -- polarizable and not."""
    atoms, basis, opts = util.load_fixture(name)
    for polar in (0, 1):
        o = dict(opts) if polar else dict(opts, polarization=0, polar_iterative=0, polar_ewald=0)
        rng = np.random.default_rng(11 + polar)
        cur = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in atoms.items()}
        S = energy.System(cur, basis, o)
        S.energy()
        n_per = len(cur["mol_id"])
        steps = 45 if not polar else 12
        for step in range(steps):
            ids = cur["mol_id"]
            starts = [0] + [i for i in range(1, len(ids)) if ids[i] != ids[i - 1]] + [len(ids)]
            k = int(rng.integers(len(starts) - 1))
            a, b = starts[k], starts[k + 1]
            arrays = [key for key, v in cur.items() if isinstance(v, np.ndarray) and len(v) == len(ids)]
            if rng.random() < 0.5 and len(starts) > 8:  # remove molecule k
                cur = {key: (np.concatenate([v[:a], v[b:]]) if key in arrays else v) for key, v in cur.items()}
            else:  # insert a copy of molecule k, moved, in front of a random molecule
                at = starts[int(rng.integers(len(starts) - 1))]
                shift = rng.uniform(-0.5, 0.5, size=3) @ basis
                new = {}
                for key, v in cur.items():
                    if key not in arrays:
                        new[key] = v
                        continue
                    piece = v[a:b].copy()
                    if key == "pos":
                        piece = piece + shift
                    if key == "mol_id":
                        piece = np.full(b - a, int(v.max()) + 1, dtype=v.dtype)
                    new[key] = np.concatenate([v[:at], piece, v[at:]])
                cur = new
            S.set_atoms(cur)
            e = S.energy()
            F = energy.System(cur, basis, o)
            ef = F.energy()
            if np.isfinite(ef):
                assert abs(e - ef) <= 1e-11 * max(abs(ef), 1.0), (polar, step, e, ef)
                for key in ("n_lj_in_cutoff", "n_es_in_cutoff", "n_intra", "n_rd_excluded", "n_es_excluded", "n_frozen", "n_pairs"):
                    assert S.observables[key] == F.observables[key], (polar, step, key)
                if polar:
                    assert abs(S.observables["polarization_energy"] - F.observables["polarization_energy"]) <= 1e-10 * max(abs(ef), 1.0)
            F.close()
        import ctypes

        L = energy.lib()
        L.mpmc_debug_upload_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
        cnt = (ctypes.c_longlong * 2)()
        assert L.mpmc_debug_upload_counts(S.handle, cnt) == 0
        carried, sorted_ = int(cnt[0]), int(cnt[1])
        assert carried + sorted_ == steps + 1, (carried, sorted_)
        # not vacuous: most moves carried the order, and (long sequences) the 64-edit trigger forced real sorts in between
        assert carried >= steps // 2, (carried, sorted_)
        if steps >= 40:
            assert sorted_ >= 2, (carried, sorted_)
        S.close()
        assert n_per > 128  # (systems of <= 128 atoms keep the identity order: nothing to carry)
