import re, sys
rows={}; cur=None; order=[]
for l in open(sys.argv[1]):
    if l.startswith("=="):
        cur=l.split()[1]
        if cur not in order: order.append(cur)
        continue
    m=re.match(r"(\w+)\s+(\d+) (step|raw)\s+([\d.]+) us(?:\s+one-launch\s+([\d.]+))?", l)
    if m: rows.setdefault((m.group(1),int(m.group(2))),{}).setdefault(cur,[]).append((float(m.group(4)), float(m.group(5) or 0)))
print("%-16s"%"case", "  ".join("%-22s"%o.split("/")[-1][:22] for o in order))
for k,v in rows.items():
    base=min(x[0] for x in v[order[0]]); baser=min(x[1] for x in v[order[0]])
    out=[]
    for o in order:
        s=min(x[0] for x in v[o]); r=min(x[1] for x in v[o])
        out.append("%7.2f(%+5.1f%%) %6.2f(%+5.1f%%)"%(s,100*(s/base-1),r,100*(r/baser-1) if baser else 0))
    print("%-7s %8d"%k, "  ".join(out))
