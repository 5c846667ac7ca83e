#!/usr/bin/env python3
"""Build-time guard for kernels whose synchronisation COUNTS vector-memory operations (`s_waitcnt vmcnt(N)` with N = "the stores this
wave issued behind its LDS-DMA"): a register spill adds scratch loads / stores to that queue and the count would let a wave read an LDS
stage before its DMA has landed -- silently wrong results, not a crash.  Reads hipcc's -Rpass-analysis=kernel-resource-usage remarks
and fails if a kernel matching one of the patterns uses scratch.

usage: check_scratch.py <remarks file> <regex> [<regex> ...]"""
import re
import sys


def main():
    path, pats = sys.argv[1], [re.compile(p) for p in sys.argv[2:]]
    name, bad, seen = None, [], 0
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and any(p.search(name) for p in pats):
            seen += 1
            if int(m.group(1)) != 0:
                bad.append((name, int(m.group(1))))
    if not seen:
        sys.exit("check_scratch: no kernel in %s matches %s" % (path, [p.pattern for p in pats]))
    if bad:
        for n, b in bad:
            print("check_scratch: %s spills %d bytes per lane: its counted s_waitcnt vmcnt(N) is no longer valid" % (n, b), file=sys.stderr)
        sys.exit(1)
    print("check_scratch: %d kernel instantiations with counted waits, none uses scratch" % seen)


if __name__ == "__main__":
    main()
