"""Generated-code check of csrc/batchteam.hip.h (build tooling; used by build() and tests/test_host.py).

batch_team_kernel issues its loads as inline assembly -- invisible to the compiler's wait-count bookkeeping --
and covers them with explicit `s_waitcnt vmcnt(n)`.  The compiler therefore believes the destination
registers are valid at once: a copy, a spill or any other use of one of them between the load and the
wait that covers it would silently read stale bits.  Whether the compiler did any of that can only be
seen in the code it generated, so the build checks THAT code, with the hipcc that built the library:

* nothing is spilled (ScratchSize 0);
* from the moment a register of the G tiles (three sets of 4 patches x 2 x 16 bytes = 96 VGPRs) has
  received its first load -- the prologue's requests included -- until the loop's last barrier, it is touched
  by nothing but the loads themselves, the MFMAs and the 16-byte LDS stores that park it;
* the destination of a small exchange load (8 bytes) is not touched before a vmcnt wait that covers it:
  `s_waitcnt vmcnt(n)` covers a load once at least n vector-memory instructions were issued behind it
  (gfx9 counts loads and stores in the same counter, in order).

`check()` returns the list of findings (empty: clean).  `stamp()` records the verdict next to the
library together with the compiler's version; `_lib.load()` switches the team form off
(GRAVHMC_BATCH_TEAM=0: the two-pass kernels need no hand-counted waits) when the stamp is missing or
lists findings (the stamp is written by the build that wrote the library, with that build's compiler).
"""
import json
import os
import re
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
STAMP = os.path.join(_HERE, "libgravhmc.isa.json")
HEADERS = ("kernels.hip.h", "batch.hip.h", "resident.hip.h", "mfbatch.hip.h", "batchteam.hip.h")
SYMBOL = "_ZN3ghk17batch_team_kernelENS_12BatchAdjArgsENS_6BtArgsE"
_VMEM = re.compile(r"(global|buffer|flat|scratch)_(load|store|atomic)")


def hipcc_version(hipcc):
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError:
        return "unknown"
    return " | ".join(l.strip() for l in out.splitlines() if l.strip())[:400]


def _regs(code):
    found = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", code):
        found.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", code):
        found.add(int(m.group(1)))
    return found


def scan(asm_text, report_text):
    """Findings in the generated code of batch_team_kernel (list of strings)."""
    bad = []
    rep = report_text[report_text.index("batch_team_kernel"):]
    rep = rep[:rep.index("LDS Size")]
    if not (re.search(r"ScratchSize \[bytes/lane\]: 0\b", rep) and re.search(r"VGPRs Spill: 0\b", rep)):
        bad.append("spills: " + " ".join(rep.split())[:300])
    body = asm_text[asm_text.index(SYMBOL + ":"):]
    body = body[:body.index(".Lfunc_end")]
    lines = body.split("\n")
    start = next(i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l)
    # (the loop's barriers are bt_lds_barrier()'s inline assembly; the code behind the loop has __syncthreads())
    last_barrier = max(i for i, l in enumerate(lines)
                       if re.match(r"\s*s_barrier", l) and any("#ASMSTART" in x for x in lines[i - 3:i]))
    code = [(i, l.split(";")[0].strip()) for i, l in enumerate(lines)]
    code = [(i, c) for i, c in code if c and not c.startswith(".")]
    tile_load = re.compile(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off.* nt")
    tile = set()
    for i, c in code:
        m = tile_load.match(c)
        if m and i >= start:
            tile.update(range(int(m.group(1)), int(m.group(2)) + 1))
    if len(tile) != 96:
        bad.append("G tile registers: %d, expected 96" % len(tile))
    allowed = ("v_mfma_f64_16x16x4_f64", "global_load_dwordx4", "ds_write_b128")
    # Tile registers that have received a load so far.  The prologue is straight-line code with forward
    # branches only (a tile that exists is requested, one that does not is zeroed on the other arm): the
    # state at a label is the union over the jumps that reach it and the fall-through; inside the loop
    # every tile register counts as in flight all the time.
    loaded, at_label = set(), {}
    for i, raw in enumerate(lines):
        if i > last_barrier:
            break
        lab = re.match(r"(\.LBB\d+_\d+):", raw.strip())
        if lab and i < start:
            loaded = (loaded or set()) | at_label.get(lab.group(1), set())
            continue
        c = raw.split(";")[0].strip()
        if not c or c.startswith("."):
            continue
        if i >= start:
            loaded = tile
        if loaded is None:       # behind an unconditional branch, before the next label: not reachable
            continue
        op = c.split()[0]
        if i < start and (op == "s_branch" or op.startswith("s_cbranch")):
            tgt = c.split()[-1]
            at_label[tgt] = at_label.get(tgt, set()) | loaded
            if op == "s_branch":
                loaded = None
            continue
        m = tile_load.match(c)
        if m:
            loaded = loaded | (set(range(int(m.group(1)), int(m.group(2)) + 1)) & tile)
            continue
        if _regs(c) & loaded and op not in allowed:
            bad.append("line %d touches a G tile register with a load in flight: %s" % (i, c))
    # the small loads of the exchange (8-byte loads): nothing reads or writes their destination before a wait
    # that covers them
    pending = {}         # register -> vector-memory instructions issued behind its load
    for i, c in code:
        if i < start or i > last_barrier:
            continue
        op = c.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", c)
            if m:
                n = int(m.group(1))
                pending = {r: k for r, k in pending.items() if k < n}
            continue
        m = re.match(r"global_load_dwordx2 v\[(\d+):(\d+)\]", c)
        if not m and _regs(c) & set(pending):
            bad.append("line %d touches the destination of an exchange load before its wait: %s" % (i, c))
        if _VMEM.match(op):
            pending = {r: k + 1 for r, k in pending.items()}
        if m:
            pending.update({r: 0 for r in range(int(m.group(1)), int(m.group(2)) + 1)})
    return bad


def check(hipcc, workdir=None):
    """Compile the kernels' headers to gfx950 assembly with `hipcc` and scan batch_team_kernel."""
    own = workdir is None
    d = tempfile.mkdtemp(prefix="gravhmc_isa_") if own else str(workdir)
    try:
        tu, asm = os.path.join(d, "bt.hip"), os.path.join(d, "bt.s")
        with open(tu, "w") as f:
            f.write("".join('#include "%s"\n' % os.path.join(CSRC, h) for h in HEADERS))
        out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", tu,
                              "-o", asm, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if out.returncode != 0:
            return ["hipcc -S failed: " + out.stderr[-1500:]]
        try:
            return scan(open(asm).read(), out.stderr)
        except (ValueError, StopIteration) as e:   # the code no longer has the shape the scan knows
            return ["scan could not find its landmarks in the generated code: %r" % (e,)]
    finally:
        if own:
            import shutil
            shutil.rmtree(d, ignore_errors=True)


def stamp(hipcc):
    """Run the check and record {compiler, findings} next to the library.  Returns the findings."""
    bad = check(hipcc)
    with open(STAMP, "w") as f:
        json.dump({"hipcc": hipcc_version(hipcc), "findings": bad}, f, indent=1)
    return bad


def team_form_cleared():
    """True when the library next to this file was built by a compiler whose batch_team_kernel code
    passed the scan (what _lib.load() asks before it leaves the team form enabled)."""
    try:
        doc = json.load(open(STAMP))
    except (OSError, ValueError):
        return False
    return doc.get("findings") == []
