"""The dipole solvers on the 10 000-atom benchmark box: wall time per evaluation and HIP-event time of the solver's kernels
(python tools/solver_time.py).  compact = production path (16 B/pair store + far-field recompute), matrix_free = every tensor
recomputed, dense = the reference's 3N x 3N matrix in device memory with the contraction on the fp64 matrix cores, gauss_seidel =
`polar_gs on` (in-place sweeps in atom order)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from mpmcxx_amd import energy
atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
n3 = 3 * ((10000 + 63) // 64 * 64)
nt = n3 // 192
for label, extra in (("compact", {"solver": "compact"}), ("matrix_free", {"solver": "matrix_free"}), ("dense", {"solver": "dense"}),
                     ("dense_whole_matrix", {"solver": "dense"}), ("gauss_seidel", {"polar_gs": 1})):
    o = dict(opts); o.update(extra)
    energy.configure("dense_symmetric", 0 if label == "dense_whole_matrix" else 1)  # (rounds 1-3 read all of A; round 4 its upper block triangle)
    try:
        S = energy.System(atoms, basis, o)
    finally:
        energy.configure("dense_symmetric", 1)
    e = S.energy()
    S.set_profiling(True); S.timings(reset=True)
    reps = 1 if label == "gauss_seidel" else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        e = S.energy()
    dt = (time.perf_counter() - t0) / reps
    t = S.timings(reset=True)
    it = t["dipole_iter"]; ten = t["tensor"]
    line = (f"{label:12s}: {dt*1e3:8.2f} ms per evaluation ({1/dt:7.1f}/s), pol {S.observables['polarization_energy']:.12e}, "
            f"iteration kernel {it['ms']/max(it['launches'],1):.4f} ms x {it['launches']//reps}")
    if ten["launches"]:
        line += f", matrix build {ten['ms']/ten['launches']:.3f} ms"
    if label.startswith("dense"):
        sec = it['ms'] / it['launches'] * 1e-3
        full = n3 * n3 * 8 / 1e9
        if label == "dense":  # tile pairs I <= J of 192 x 192 doubles; off-diagonal blocks feed two MFMA products, diagonal ones one
            gb = nt * (nt + 1) // 2 * 192 * 192 * 8 / 1e9
            mfma = 2.0 * 16 * 192 * 192 * (nt * (nt - 1) + nt) / sec / 1e12
            line += (f"; symmetric contraction reads {gb:.2f} GB of the {full:.2f} GB matrix -> {gb / sec / 1e3:.2f} TB/s from HBM, {full / sec / 1e3:.2f} TB/s in "
                     f"terms of the whole matrix, {mfma:.1f} TFLOP/s issued on v_mfma_f64 (1/16 useful)")
        else:
            line += f"; dense contraction reads {full:.2f} GB -> {full / sec / 1e3:.2f} TB/s, {2*n3*n3*16/sec/1e12:.1f} TFLOP/s issued on v_mfma_f64 (1/16 useful)"
    tot, tens = S.memory_usage()
    print(line + f"; device memory {tot/2**30:.2f} GiB", flush=True)
    S.close()
