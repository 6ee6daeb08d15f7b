set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; rm -rf $OUT/klt_dump; mkdir -p $OUT/klt_dump; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
cd /tmp
PVIO_KLT_DUMP=/tmp/da PVIO_SEQ_IMAGE=oracle timeout 900 python $R/tests/chain_run.py $R/oracle/_ref/libpvio_ref.so /tmp/ra 360 8 3 25.0 full_relief_sweep 2>&1 | tail -1
PVIO_KLT_DUMP=/tmp/db PVIO_SEQ_IMAGE=hip timeout 900 python $R/tests/chain_run.py $R/oracle/_ref/libpvio_dropin.so /tmp/rb 360 8 3 25.0 full_relief_sweep 2>&1 | grep -v "$F" | tail -1
python - <<'PY'
import numpy as np
def load(path, kmax):
    b=open(path,'rb').read(); o=0; out=[]
    while o<len(b) and len(out)<kmax:
        n=int(np.frombuffer(b,np.int32,1,o)[0]); o+=4
        r=[np.frombuffer(b,np.float32,2*n,o+8*n*j).reshape(n,2).copy() for j in range(3)]; o+=24*n
        st=np.frombuffer(b,np.uint8,n,o).copy(); o+=n
        out.append((r[0],r[1],r[2],st))
    return out
A=load('/tmp/da_oracle.bin',70); B=load('/tmp/db_hip.bin',70)
import pickle
pickle.dump((A,B),open('/root/repo/gpurun_out/klt_dump/first70.pkl','wb'))
for k,(a,b) in enumerate(zip(A,B)):
    if a[0].shape!=b[0].shape: print('call',k,'shapes differ',a[0].shape,b[0].shape); break
    sp=(a[0]==b[0]).all(); si=(a[1].view(np.int32)==b[1].view(np.int32)).all(); ss=(a[3]==b[3]).all(); ok=(a[3]>0)&(b[3]>0); so=(a[2][ok]==b[2][ok]).all()
    if not (sp and si and ss and so):
        print('call',k,'prev same',sp,'init same',si,'status same',ss,'out same',so)
        di=np.where((a[1]!=b[1]).any(1))[0]; print(' init differs at',di,a[1][di],b[1][di])
        do=np.where(((a[2]!=b[2]).any(1)&ok)|(a[3]!=b[3]))[0]; print(' out differs at',do,'prev',a[0][do],'init',a[1][do],b[1][do],'out',a[2][do],b[2][do],'status',a[3][do],b[3][do])
        break
PY
