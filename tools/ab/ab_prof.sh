(cd _r1 && python bench.py --no-cpu-baseline --steps 10 --profile-out ../gpurun_out/ab_r1_prof.json > /dev/null 2>&1)
PLYOLO_PW=0 python bench.py --no-cpu-baseline --steps 10 --profile-out gpurun_out/ab_pw0_prof.json > /dev/null 2>&1
PLYOLO_PW=1 python bench.py --no-cpu-baseline --steps 10 --profile-out gpurun_out/ab_pw1_prof.json > /dev/null 2>&1
python - <<'PY'
import json
def fam(f):
    d=json.load(open(f)); out={}
    for r in d['rows']:
        k=r[0].split('<')[0]; out.setdefault(k,[0,0.0]); out[k][0]+=r[1]; out[k][1]+=r[2]
    return out, d['sum_ms']
a,sa=fam('gpurun_out/ab_r1_prof.json'); b,sb=fam('gpurun_out/ab_pw0_prof.json'); c,sc=fam('gpurun_out/ab_pw1_prof.json')
print("sum r1 %.3f pw0 %.3f pw1 %.3f"%(sa,sb,sc))
for k in sorted(set(a)|set(b)|set(c), key=lambda k:-(a.get(k,[0,0])[1]+b.get(k,[0,0])[1])):
    print("%-22s r1 %3d %7.3f | pw0 %3d %7.3f | pw1 %3d %7.3f"%(k,a.get(k,[0,0])[0],a.get(k,[0,0])[1],b.get(k,[0,0])[0],b.get(k,[0,0])[1],c.get(k,[0,0])[0],c.get(k,[0,0])[1]))
PY
