for v in trace tracens traceall; do cp build_abl/lib_$v.so neuralcodecs_amd/libnc_mi355x.so; echo "== $v"; python tools/probe/convtrace.py 384 5568 3 | tail -1; python - <<PY
import numpy as np
b=np.load("gpurun_out/convtrace.npy"); st=b[:,3:].astype(np.int64)
t0=np.array([r[r>0].min() for r in st]); t1=np.array([r.max() for r in st])
order=np.argsort(t0); 
# split the two launches at the largest gap between consecutive block starts
s=np.sort(t0); gaps=np.diff(s); k=np.argmax(gaps); cut=(s[k]+s[k+1])//2
for name,mask in (("launch1", t0<cut),("launch2", t0>=cut)):
    print(name, "span ticks", int(t1[mask].max()-t0[mask].min()), "waves", int(mask.sum()))
PY
done
