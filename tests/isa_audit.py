"""Build-time ISA audit of the shipped gfx950 code objects: pending-load hazards.

hipcc does not count the memory operations of an ``asm`` statement: a VGPR that an inline-asm ``global_load`` / ``ds_read`` will fill
counts as written when the statement ends, so the compiler may read, copy, spill or re-use it before the data lands (the cause of the
round-3 ``Memory access fault`` in an experimental attention kernel: profiles/r03_attention_lazy.txt section 4a).  The rule this module
checks on the DISASSEMBLY of every kernel in ``lib/libleclip_hip.so`` - compiler-counted and hand-counted loads alike, since the
disassembly cannot tell them apart and both must obey it:

    between the issue of a load with a VGPR / AGPR destination and the ``s_waitcnt`` that retires it, no other instruction reads or
    writes any of its destination registers.

Counter model (gfx950): ``vmcnt`` counts vector-memory loads, stores, atomics and LDS-DMA in issue order; ``lgkmcnt`` counts DS and
scalar-memory operations (DS operations return in order; scalar loads have no vector destination and only occupy a slot here).
``s_waitcnt <counter>(N)`` retires all but the N youngest operations of that counter.  The walk follows the control-flow graph
(branch targets from the disassembly) to a fixed point over (vm queue, lgkm queue) states, so a load pending across a loop back
edge is seen at the loop head too.

Test infrastructure only (tests/test_host_logic.py); needs llvm-objdump from the ROCm toolchain, no GPU.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile
from typing import Dict, List, Tuple

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_LINE = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
_TARGET = re.compile(r"<[^>]*\+0x([0-9a-f]+)>\s*$")
_WAIT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


def code_objects(lib_path: str, workdir: str) -> List[str]:
    """Unbundle the gfx950 code objects of a HIP shared library into ``workdir`` (llvm-objdump writes them next to its input)."""
    local = os.path.join(workdir, os.path.basename(lib_path))
    shutil.copy(lib_path, local)
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f and "gfx950" in f)


def disassemble(code_object: str) -> Dict[str, List[Tuple[int, str, str, int]]]:
    """{kernel: [(address, mnemonic, operands, branch target address or -1)]}"""
    out = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", code_object], check=True, capture_output=True, text=True).stdout
    funcs: Dict[str, List[Tuple[int, str, str, int]]] = {}
    cur, start = None, 0
    for line in out.splitlines():
        m = _FUNC.match(line)
        if m:
            start = int(m.group(1), 16)
            cur = funcs.setdefault(m.group(2), [])
            continue
        m = _LINE.match(line)
        if m is None or cur is None:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = -1
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = _TARGET.search(line)
            if t:
                tgt = start + int(t.group(1), 16)
        cur.append((addr, mnem, ops, tgt))
    return funcs


def _regs(text: str) -> frozenset:
    s = set()
    for m in _REG.finditer(text):
        if m.group(1):
            s.add((m.group(1), int(m.group(2))))
        else:
            s.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return frozenset(s)


def _classify(mnem: str, ops: str):
    """-> (counter or None, destination registers).  LDS-DMA, stores and atomics without return occupy a slot with no destination."""
    if mnem.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "global_atomic", "buffer_atomic", "flat_atomic")):
        if "lds" in mnem or re.search(r"\blds\b", ops):
            return "vm", frozenset()
        first = ops.split(",")[0]
        dest = _regs(first) if (mnem.find("atomic") < 0 or "sc0" in ops or "glc" in ops) else frozenset()
        return "vm", dest
    if mnem.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "buffer_wbl2", "buffer_inv")):
        return "vm", frozenset()
    if mnem.startswith("ds_"):
        if mnem.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in mnem:
            return "lgkm", _regs(ops.split(",")[0])
        return "lgkm", frozenset()
    if mnem.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_sendmsg", "s_atc", "s_dcache", "s_scratch_load", "s_store")):
        return "lgkm", frozenset()
    return None, frozenset()


def _trim(queue: tuple, keep: int) -> tuple:
    """Keep the ``keep`` youngest operations.  Queue items, oldest first: ``(issue address, destination registers)`` for a load with a
    vector destination, or an int = a run of that many operations without one (stores, LDS-DMA, DS writes, scalar loads: they only
    occupy counter slots).  A leading run is dropped: operations older than the oldest pending destination cannot matter."""
    out, left = [], keep
    for item in reversed(queue):
        if left <= 0:
            break
        if isinstance(item, int):
            take = item if item <= left else left
            out.append(take)
            left -= take
        else:
            out.append(item)
            left -= 1
    while out and isinstance(out[-1], int):
        out.pop()
    out.reverse()
    return tuple(out)


def _push(queue: tuple, entry, cap: int) -> tuple:
    if entry is None:                                   # no destination: extend (or start) the trailing run - unless nothing is pending
        if not queue:
            return queue
        if isinstance(queue[-1], int):
            return queue[:-1] + (min(queue[-1] + 1, cap),)
        return queue + (1,)
    return _trim(queue + (entry,), cap)


def audit_kernel(name: str, insts: List[Tuple[int, str, str, int]], max_states: int = 3000000) -> List[str]:
    """Violations (strings) of the pending-destination rule in one kernel."""
    index = {a: i for i, (a, _, _, _) in enumerate(insts)}
    # per instruction, once: registers named, counter class + destination, wait fields
    pre = []
    for addr, mnem, ops, tgt in insts:
        used = _regs(ops)
        waits = [(c, int(n)) for c, n in _WAIT.findall(ops)] if mnem == "s_waitcnt" else None
        which, dest = (None, frozenset()) if waits is not None else _classify(mnem, ops)
        pre.append((used, waits, which, dest, index.get(tgt, -1) if tgt >= 0 else -1, mnem))
    seen = set()
    work = [(0, (), ())]
    bad: Dict[Tuple[int, int], str] = {}
    steps = 0
    while work:
        pc, vm, lg = work.pop()
        while pc < len(insts):
            key = (pc, vm, lg)
            if key in seen:
                break
            seen.add(key)
            steps += 1
            if steps > max_states:
                return [f"{name}: state budget exceeded (audit inconclusive)"]
            used, waits, which, dest, tpc, mnem = pre[pc]
            if used and (vm or lg):
                for qi, q in enumerate((vm, lg)):
                    for k, item in enumerate(q):
                        if isinstance(item, int) or item[1].isdisjoint(used):
                            continue
                        own = "vm" if qi == 0 else "lgkm"
                        if which == own and dest and (used & item[1]) <= dest and not (_regs(insts[pc][2].split(",", 1)[1] if "," in insts[pc][2] else "") & item[1]):
                            # a younger load of the SAME counter overwrites the destination: loads of one counter return in order, the
                            # younger one lands last - legal, and the older one no longer owns those registers
                            rest = item[1] - dest
                            q = q[:k] + ((((item[0], rest),) if rest else (1,))) + q[k + 1:]
                            if qi == 0:
                                vm = q
                            else:
                                lg = q
                            continue
                        hit = sorted(item[1] & used)[0]
                        a = insts[pc][0]
                        bad.setdefault((item[0], a), f"{name}: {mnem} {insts[pc][2]} @0x{a:x} touches {hit[0]}{hit[1]}, destination of the load "
                                       f"issued @0x{item[0]:x} and not yet retired")
            if waits is not None:
                for cnt, n in waits:
                    if cnt == "vmcnt":
                        vm = _trim(vm, n)
                    elif cnt == "lgkmcnt":
                        lg = _trim(lg, n)
            elif which == "vm":
                vm = _push(vm, (insts[pc][0], dest) if dest else None, 64)
            elif which == "lgkm":
                lg = _push(lg, (insts[pc][0], dest) if dest else None, 16)
            if mnem == "s_endpgm":
                break
            if tpc >= 0:
                if mnem == "s_branch":
                    pc = tpc
                    continue
                work.append((tpc, vm, lg))
            elif mnem in ("s_setpc_b64", "s_swappc_b64"):
                break
            pc += 1
    return sorted(bad.values())


def audit_library(lib_path: str, only=None, skip=()) -> Tuple[int, List[str]]:
    """(kernels audited, violations) over every gfx950 kernel of the library (``only``: substring filter on the mangled name;
    ``skip``: substrings of kernels left out)."""
    with tempfile.TemporaryDirectory() as tmp:
        n, out = 0, []
        for co in code_objects(lib_path, tmp):
            for name, insts in disassemble(co).items():
                if (only and only not in name) or any(k in name for k in skip):
                    continue
                n += 1
                out += audit_kernel(name, insts)
        return n, out


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        here, "..", "language-enhanced-clip-for-multi-label-image-recognition_amd", "lib", "libleclip_hip.so")
    count, problems = audit_library(lib, sys.argv[2] if len(sys.argv) > 2 else None)
    print(f"{count} kernels audited, {len(problems)} violations")
    for p in problems[:50]:
        print(" ", p)
    sys.exit(1 if problems else 0)
