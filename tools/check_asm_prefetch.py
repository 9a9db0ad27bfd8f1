#!/usr/bin/env python3
"""Build-time check of sort.hip's software prefetch (SortPartWaveKernel<8, true>).

The kernel fetches the NEXT part's records with `asm volatile("global_load_dwordx4 ...")` while the current part
runs through the sorting network, and waits for them at the top of the next iteration (WaitRecs).  The compiler
does not track loads issued by inline assembly: it considers the destination registers defined at the asm
statement, so a copy, spill or AGPR move of them that register allocation places between the load and the wait
would read registers the load has not filled yet -- silently, and possibly only with another compiler version or
other flags.  This script disassembles the kernel as built and proves that this did not happen:

  every vector-memory load is tracked from its issue to the `s_waitcnt` that retires it (vmcnt is an in-order
  counter: `vmcnt(N)` leaves the newest N operations outstanding), through the loop's back edges; any instruction
  that names a destination register of a load still outstanding fails the build.

Usage: check_asm_prefetch.py <device assembly (.s)> [kernel-name-substring]
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM = re.compile(r"^\s*(global|buffer|flat|scratch)_(load|store|atomic)\w*\s")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)$")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*s_(?:c?branch)\w*\s+(\.LBB\d+_\d+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernel_body(lines, name):
    start = next((i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(";")[0].rstrip().endswith(":")), None)
    if start is None:
        raise SystemExit(f"check_asm_prefetch: kernel {name} not found")
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start + 1:end + 1]


def check(body, name):
    code = [l.split(";")[0].rstrip() for l in body]
    labels = {m.group(1): i for i, l in enumerate(code) if (m := LABEL.match(l))}
    # abstract state at an instruction = tuple of outstanding vector-memory ops, oldest first; each op = frozenset of the
    # VGPRs a load will write (empty for stores).  Explore all paths (worklist over (pc, state)), states are small.
    seen = set()
    work = [(0, ())]
    violations = []
    n_loads = 0
    while work:
        pc, st = work.pop()
        while pc < len(code):
            key = (pc, st)
            if key in seen:
                break
            seen.add(key)
            line = code[pc].strip()
            if not line or line.startswith(".") and not LABEL.match(code[pc]) or LABEL.match(code[pc]):
                pc += 1
                continue
            w = WAIT.match(line)
            if w:
                m = re.search(r"vmcnt\((\d+)\)", w.group(1))
                if m:
                    keep = int(m.group(1))
                    st = st[len(st) - keep:] if keep < len(st) else st
                    if keep == 0:
                        st = ()
                elif not re.search(r"lgkmcnt|expcnt", w.group(1)):  # a raw immediate: treat as a full wait only if 0
                    imm = w.group(1).strip()
                    if imm in ("0", "0x0"):
                        st = ()
                pc += 1
                continue
            used = regs_of(line)
            pending = set().union(*st) if st else set()
            if used & pending:
                violations.append((pc, line, sorted(used & pending)))
            v = VMEM.match(line)
            if v:
                dst = frozenset()
                if v.group(2) == "load" or (v.group(2) == "atomic" and " glc" in line):
                    first = line.split(None, 1)[1].split(",")[0]
                    dst = frozenset(regs_of(first))
                    n_loads += 1
                st = st + (dst,)
                if len(st) > 64:
                    st = st[-64:]
            b = BRANCH.match(line)
            if b and b.group(1) in labels:
                work.append((labels[b.group(1)], st))
                if line.startswith("s_branch"):
                    break
            if "s_endpgm" in line:
                break
            pc += 1
    return violations, n_loads


def main():
    path = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "SortPartWaveKernelILi8ELb1E"
    lines = open(path).read().splitlines()
    body = kernel_body(lines, name)
    asm_loads = sum(1 for l in body if "global_load_dwordx4" in l)
    violations, n_loads = check(body, name)
    if asm_loads < 16:
        raise SystemExit(f"check_asm_prefetch: expected the two prefetch groups (16 global_load_dwordx4) in {name}, found {asm_loads}")
    if violations:
        for pc, line, regs in violations[:10]:
            print(f"check_asm_prefetch: {name}: instruction {pc}: `{line}` touches v{regs} while a load into them is outstanding",
                  file=sys.stderr)
        raise SystemExit(1)
    print(f"check_asm_prefetch: {name}: {asm_loads} dwordx4 loads, no register of an outstanding load is touched before its wait")


if __name__ == "__main__":
    main()
