import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle.pyoracle import Oracle
from ygz_slam_amd import _lib, synth
from test_gpu_parity import _frames
o = Oracle()
imgs, _, _ = _frames(2, 640, 480, seed=2)
ctx = _lib.HipContext(max_frames=2)
ctx.upload_gray(0, imgs[0]); ctx.build_pyramid(0,1); ctx.detect(0,1)
kp = ctx.get_keypoints(0)
ok = o.detect(o.pyramid(imgs[0],3))
bad = np.nonzero(kp['score'] != ok['score'])[0]
print(len(bad), len(ok))
for i in bad[:10]:
    print(i, kp['px'][i], kp['level'][i], repr(kp['score'][i]), repr(ok['score'][i]), kp['score'][i].view(np.uint32) if hasattr(kp['score'][i],'view') else '', )
    L = int(ok['level'][i]); lv = o.pyramid(imgs[0],3)[L]
    x = int(ok['px'][i])>>L; y = int(ok['py'][i])>>L
    w = lv[y-5:y+5, x-5:x+5].astype(np.int64)
    dXX=dYY=dXY=0
    for yy in range(y-4,y+4):
        for xx in range(x-4,x+4):
            dx = int(lv[yy,xx+1])-int(lv[yy,xx-1]); dy=int(lv[yy+1,xx])-int(lv[yy-1,xx])
            dXX+=dx*dx; dYY+=dy*dy; dXY+=dx*dy
    a=np.float32(dXX/128.0); b=np.float32(dYY/128.0); c=np.float32(dXY/128.0)
    tr=np.float32(a+b); disc=np.float32(np.float32(tr*tr)-np.float32(np.float32(4)*np.float32(np.float32(a*b)-np.float32(c*c))))
    print("   numpy:", dXX,dYY,dXY, repr(np.float32(0.5)*np.float32(tr-np.sqrt(disc))), repr(disc))
