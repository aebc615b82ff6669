import re,sys
best={}
cur=None
for ln in open(sys.argv[1]):
    m=re.match(r"(enc\d|dec\d) fwd",ln)
    if m: cur=m.group(1); continue
    m=re.match(r"\s+mode (\d) split\s+(\d+): gemm\s+([\d.]+) us\s+gemm\+reduce\s+([\d.]+)",ln)
    if m and cur:
        md,sp,g,gr=int(m.group(1)),int(m.group(2)),float(m.group(3)),float(m.group(4))
        k=(cur,md)
        if k not in best or gr<best[k][1]: best[k]=(sp,gr,g)
tot={0:0,1:0,2:0,3:0,"best":0}
for l in ["enc0","enc1","enc2","enc3","dec0","dec1","dec2","dec3"]:
    row=[best.get((l,md)) for md in (0,1,2,3)]
    print(l, row)
    for md in (0,1,2,3):
        if row[md]: tot[md]+=row[md][1]
    tot["best"]+=min(r[1] for r in row if r)
print(tot)
