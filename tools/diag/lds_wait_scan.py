"""Static scan of the kernels' ISA for the LDS pattern that round 6 found to be unsafe beside another kernel's ds_read_b64_tr_b16:
a COUNTED `s_waitcnt lgkmcnt(N > 0)` that retires a READ while an LDS operation of the OTHER banking class stays in flight (4-byte class:
ds_read/write_b32, _b16, _b8, ds_read2_b32, LDS atomics; 8/16-byte class: ds_read/write_b64 / b96 / b128, ds_read2_b64, tr reads).  mt_token_mha_fwd's value sweep
([6 x b128, 4 x read2_b32, 2 x b128] + lgkmcnt(5)) returned stale 16-byte results in lanes 48-63 once in ~10 launches beside
mt_gemm_tn_f16 on another stream; the same instructions behind ONE lgkmcnt(0) never did (tools/diag/victim_stress2.py).
    for f in modaltune_amd/csrc/*.hip; do hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/isa/$(basename $f .hip).s $f; done
    python tools/diag/lds_wait_scan.py /tmp/isa/*.s"""
import re
import sys

SMALL = re.compile(r"\bds_(read|write)(2|2st64)?_(b32|b8|b16|u8|i8|u16|i16|u8_d16|u8_d16_hi|u16_d16|u16_d16_hi|b8_d16_hi|b16_d16_hi)\b"
                   r"|\bds_(add|sub|max|min|and|or|xor|inc|dec|cmpst|wrxchg)\w*\b|\bds_write_addtid_b32\b|\bds_read_addtid_b32\b")
WIDE = re.compile(r"\bds_(read|write)(2|2st64)?_(b64|b96|b128)\b|\bds_read_b64_tr_b\d+\b|\bds_read_b96_tr_b6\b")
ANYLDS = re.compile(r"^\s*ds_")
WAIT = re.compile(r"s_waitcnt\b(.*)")


def scan(path):
    """{kernel symbol: number of unsafe counted lgkmcnt waits}.  In-order model: `lgkmcnt(n)` passes when all but the n youngest LDS
    operations are done.  If operations of different banking classes can complete out of order (what round 6 observed), the wait is
    unsafe when it RETIRES a read (an older operation whose result is about to be used) while one of the n operations it leaves in
    flight is of the OTHER class -- reads, writes and LDS atomics all count as operations; ds_bpermute / ds_swizzle touch no memory bank
    and are ignored.  What is in flight stays in flight across basic blocks (conservative)."""
    kernel, inflight, flagged = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\w+|mt_\w+|\w+_kernel\w*):", line)
        if m:
            kernel, inflight = m.group(1), []
            continue
        if kernel is None:
            continue
        if line.startswith(".Lfunc_end"):
            kernel = None
            continue
        s = line.split(";")[0]
        if ANYLDS.match(s):
            op = s.split()[0]
            cls = "small" if SMALL.search(s) else "wide" if WIDE.search(s) else None
            if cls is not None:
                inflight.append((cls, "read" in op or "rtn" in op))
            continue
        w = WAIT.search(s)
        if w:
            mm = re.search(r"lgkmcnt\((\d+)\)", w.group(1))
            if mm is None:
                continue          # (a wait that names only vmcnt leaves the LDS counter alone)
            n = int(mm.group(1))
            if 0 < n < len(inflight):
                retired, young = inflight[:len(inflight) - n], inflight[len(inflight) - n:]
                if any(is_read and any(yc != c for yc, _ in young) for c, is_read in retired):
                    flagged[kernel] = flagged.get(kernel, 0) + 1
                inflight = young
            elif n == 0:
                inflight = []
    return flagged


if __name__ == "__main__":
    for path in sys.argv[1:]:
        for k, c in sorted(scan(path).items()):
            print(f"{path.split('/')[-1]:18s} {c:4d} unsafe counted lgkmcnt waits (an older read retired while an operation of the other LDS class is still in flight) in {k}")
