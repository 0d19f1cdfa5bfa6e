import json,sys
d=json.load(open(sys.argv[1]))
for r in d['rows']:
    if any(k in r[0] for k in ('pack','loss','fill','memset','focus','bias','spp','copy','upsample')): print("%-28s x%3d %8.1f us"%(r[0],r[1],r[2]*1e3))
print(d['sum_ms'], sum(r[1] for r in d['rows']))
