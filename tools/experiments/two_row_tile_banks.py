import itertools
GROUPS=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
def conflicts(addr_of_lane):
    worst=1
    for g in GROUPS:
        cnt={}
        for l in g:
            a=addr_of_lane(l)
            quad=(a//16)%16
            cnt.setdefault(quad,set()).add(a//16)
        worst=max(worst,max(len(v) for v in cnt.values()))
    return worst
def delta(lr): return 2*lr if lr<4 else (2*(lr-4)+1 if lr<12 else 2*(lr-8))
# check the existing scheme: 16 consecutive positions
def existing(c, ks):
    def f(l):
        q,lr=l>>4,l&15
        pos=c+delta(lr)
        piece=((ks<<2)|q)^((pos>>1)&7)
        return pos*128+piece*16
    return f
print("existing:", [conflicts(existing(c,ks)) for c in range(4) for ks in range(2)])
# two-row scheme: lane lr -> (h, xi); slot B base = pitchoff (mod 256) relative to slot A
best=[]
perms8=list(itertools.permutations(range(8)))
import random
random.seed(0)
def test(hmap, ximap, pitchoff):
    w=1
    for c in range(0,10):          # start position of the 8-pixel run (8u + kx + pad), parity and swizzle phase vary
        for ks in range(2):
            def f(l):
                q,lr=l>>4,l&15
                h=hmap[lr]; pos=c+ximap[lr]
                piece=((ks<<2)|q)^((pos>>1)&7)
                return pos*128+piece*16+h*pitchoff
            w=max(w,conflicts(f))
            if w>1: return w
    return w
found=[]
# candidate families: h = lr>>3 with xi = perm(lr&7) for both halves (same or different perms), pitch offsets
for pitchoff in range(0,256,16):
    for trial in range(4000):
        pa=random.choice(perms8); pb=random.choice(perms8)
        hmap=[lr>>3 for lr in range(16)]
        ximap=[pa[lr&7] if lr<8 else pb[lr&7] for lr in range(16)]
        if test(hmap,ximap,pitchoff)==1:
            found.append((pitchoff,pa,pb)); break
print("h=lr>>3 family:", found[:5])
found=[]
for pitchoff in range(0,256,16):
    for trial in range(4000):
        # h interleaved differently: choose which 8 of the 16 lr go to row 0
        lrs=list(range(16)); random.shuffle(lrs)
        hmap=[0]*16; ximap=[0]*16
        for i,lr in enumerate(lrs):
            hmap[lr]=i>>3; ximap[lr]=i&7
        if test(hmap,ximap,pitchoff)==1:
            found.append((pitchoff,hmap,ximap)); break
print("random family:", found[:3])
pa=(4, 6, 2, 0, 5, 3, 7, 1); pb=(3, 5, 7, 1, 6, 2, 0, 4)
hmap=[lr>>3 for lr in range(16)]; ximap=[pa[lr&7] if lr<8 else pb[lr&7] for lr in range(16)]
w=1
for c in range(0,64):
    for ks in range(2):
        def f(l):
            q,lr=l>>4,l&15
            pos=c+ximap[lr]; piece=((ks<<2)|q)^((pos>>1)&7)
            return pos*128+piece*16
        w=max(w,conflicts(f))
print("verify c 0..63:", w)
