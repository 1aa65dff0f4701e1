"""CPU: source-level rules of the product tree that a run-time test cannot pin deterministically.

Stream discipline: every context owns non-blocking streams and NOTHING in the library may run on the null stream, which
is unordered against them -- a null-stream hipMemset once raced with a context's first evaluation one time in ~4000 (fixed in 9825c63).
The deterministic guard is this scan: no synchronous / null-stream HIP API and no stream-0 kernel launch anywhere under csrc/."""
import os
import re

import util

CSRC = os.path.join(util.ROOT, "mpmcxx_amd", "csrc")
FORBIDDEN = [
    r"\bhipMemset\s*\(", r"\bhipMemcpy\s*\(", r"\bhipMemcpyToSymbol\s*\(", r"\bhipMemsetD\d+\s*\(", r"\bhipMemcpyDtoH\s*\(", r"\bhipMemcpyHtoD\s*\(",
    r"\bhipDeviceSynchronize\s*\(", r"\bhipStreamSynchronize\s*\(\s*(0|nullptr|NULL)\s*\)", r"hipStreamPerThread", r"hipStreamLegacy",
]


def sources():
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".cpp", ".hip", ".h")):
            txt = open(os.path.join(CSRC, f)).read()
            txt = re.sub(r"//[^\n]*", "", txt)
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            yield f, txt


def test_no_null_stream_api_in_the_library():
    bad = []
    for f, txt in sources():
        # the one exception (ABI 6): mpmc_device_synchronize IS a device-wide fence, called by a host program around its timing bracket --
        # hipDeviceSynchronize waits for the non-blocking streams too, and nothing of the energy path runs through it
        txt = re.sub(r'extern "C" int mpmc_device_synchronize\(int device\) \{.*?\n\}', "", txt, flags=re.S)
        for pat in FORBIDDEN:
            for m in re.finditer(pat, txt):
                bad.append((f, txt[max(0, m.start() - 40):m.end() + 20].replace("\n", " ")))
    assert not bad, bad


def test_every_kernel_launch_names_a_stream():
    """hipLaunchKernelGGL(kernel, grid, block, shmem, STREAM, ...): the stream argument is a variable (st, s2, c->stream ...), never the
    null stream; triple-chevron launches carry all four launch parameters."""
    bad = []
    for f, txt in sources():
        for m in re.finditer(r"hipLaunchKernelGGL\s*\((.*?)\)\s*;", txt, flags=re.S):
            call = " ".join(m.group(1).split())
            # ..., dim3 grid, dim3 block, <shared bytes>, <stream>, args...  -- a literal 0 / nullptr in the stream slot follows the shared-memory size
            if re.search(r",\s*(0|\d+|[\w:]+)\s*,\s*(0|nullptr|NULL)\s*(,|$)", call) and re.search(r",\s*0\s*,\s*(0|nullptr|NULL)\s*(,|$)", call):
                bad.append((f, call[:160]))
        for m in re.finditer(r"<<<([^>]*)>>>", txt):
            parts = [p.strip() for p in m.group(1).split(",")]
            if len(parts) < 4 or parts[3] in ("0", "nullptr", "NULL"):
                bad.append((f, m.group(0)))
    assert not bad, bad


def test_stale_references_are_gone():
    # pair_math.h used to cite a tests/hostcheck that never existed
    assert "tests/hostcheck" not in open(os.path.join(CSRC, "pair_math.h")).read()
