"""median of the phases in RSDSFM_XFER_TRACE lines read from stdin (tools/host_boundary_probe.py's stderr)"""
import re, sys, statistics as st
rows = {}
for line in sys.stdin:
    m = re.match(r"\[xfer\] (\w+):", line)
    if not m:
        continue
    ph = dict((k.strip(), float(v)) for k, v in re.findall(r"(?<=\s)([a-z() ]+?) \+(\d+)", line.split(":", 1)[1].split("|")[0]))
    tot = float(re.search(r"total (\d+)", line).group(1))
    rows.setdefault(m.group(1), []).append((ph, tot))
skip = int(sys.argv[1]) if len(sys.argv) > 1 else 3
drop = int(sys.argv[2]) if len(sys.argv) > 2 else 3  # (the probe's last calls use fresh arrays)
for name, r in rows.items():
    r = r[skip:len(r) - drop]
    keys = [k for k in r[0][0] if k != "begin"]
    print(name, "n=%d" % len(r), " ".join("%s %.0f" % (k, st.median(x[0].get(k, 0.0) for x in r)) for k in keys), "| total median %.0f min %.0f us" % (st.median(x[1] for x in r), min(x[1] for x in r)))
