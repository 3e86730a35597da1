import numpy as np, sys, math
sys.path.insert(0,'/root/repo')
from miniengineao_amd import synth
def study(kind,W,H,cam,seed=0x1234ABCD):
    depth = synth.make(kind,W,H,seed=seed).astype(np.float64)
    fpn = cam.far/cam.near
    lin = 1.0/((fpn-1)*depth+1)
    thick=[]
    for (a,b) in [(1,0),(2,0),(3,0),(4,0),(1,1),(1,2),(1,3),(1,4),(2,2),(2,3),(2,4),(3,3)]:
        thick.append(math.sqrt(1-(a/5)**2-(b/5)**2))
    terms=[(2,0,1),(4,0,3),(1,1,4),(2,2,8),(3,3,11),(1,3,6),(2,4,10)]
    tan=1.0/cam.proj00(W,H)
    for lv in (1,2,3,4):
        low = lin[::2**lv, ::2**lv].astype(np.float16).astype(np.float64)
        lh,lw = low.shape
        sw = -(-W//2**(lv+2))
        TM = 2*tan*10/sw
        A=16
        pad = np.pad(low,A,mode='edge')
        c = low
        invd = 1.0/c
        tot_pairs=0; res={}
        allpos_texel = np.ones_like(c,bool)
        sum1_texel = np.ones_like(c,bool)
        pair_stats=[]
        for (x,y,ti) in terms:
            invT = (1/TM)/thick[ti]
            front = invT-0.5
            if y==0: offs=[(x,0),(0,x)]
            elif x==y: offs=[(-x,x),(x,x)]
            else: offs=[(x,y),(-x,y),(y,x),(-y,x)]
            term_pos = np.ones_like(c,bool)
            for (dx,dy) in offs:
                s1 = pad[A+4*dy:A+4*dy+lh, A+4*dx:A+4*dx+lw]
                s2 = pad[A-4*dy:A-4*dy+lh, A-4*dx:A-4*dx+lw]
                d1 = s1*invT*invd-front; d2 = s2*invT*invd-front
                pos = (d1>=0)&(d2>=0)
                pair_stats.append(pos)
                term_pos &= pos
                sum1_texel &= pos & (d1+d2>=1)
            res[(x,y)] = term_pos
            allpos_texel &= term_pos
        # wave blocks: 64 wide x 2 rows
        def wave_frac(m):
            hh=(lh//2)*2; ww=(lw//64)*64
            if hh==0 or ww==0: return float('nan')
            b = m[:hh,:ww].reshape(hh//2,2,ww//64,64).all(axis=(1,3))
            return b.mean()
        def blk_frac(m,bh,bw):
            hh=(lh//bh)*bh; ww=(lw//bw)*bw
            if hh==0 or ww==0: return float('nan')
            return m[:hh,:ww].reshape(hh//bh,bh,ww//bw,bw).all(axis=(1,3)).mean()
        print(f"{kind} {W}x{H} level {lv} ({lw}x{lh}) TM={TM:.4f}")
        print("  per-lane: pair>=0 %.3f  term>=0 %s  texel-all %.3f  texel-all-sum>=1 %.3f" % (
            np.mean([p.mean() for p in pair_stats]), ' '.join('%.2f'%res[k].mean() for k in res), allpos_texel.mean(), sum1_texel.mean()))
        print("  wave(64x2): pair %.3f  term %s  texel-all %.3f" % (
            np.mean([wave_frac(p) for p in pair_stats]), ' '.join('%.2f'%wave_frac(res[k]) for k in res), wave_frac(allpos_texel)))
        print("  blk 16x4: pair %.3f texel-all %.3f ; blk 8x8 pair %.3f texel-all %.3f" % (
            np.mean([blk_frac(p,4,16) for p in pair_stats]), blk_frac(allpos_texel,4,16),
            np.mean([blk_frac(p,8,8) for p in pair_stats]), blk_frac(allpos_texel,8,8)))
study("S2",3840,2160,synth.DEFAULT_CAMERA)
study("S3",1920,1080,synth.SPONZA_CAMERA)
