import itertools
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
# write groups for ds_write_b128: 8 x 8 contiguous lanes
wgroups = [list(range(8*i, 8*i+8)) for i in range(8)]
def worst_read(RB, UPL, sw):
    UR = RB//16
    KB = max(UR//(4*UPL),1)
    worst = 0
    for s in range(KB):
        for h in range(UPL):
            for g in groups:
                slots = {}
                for lane in g:
                    r, q = lane & 15, lane >> 4
                    unit = (4*s+q)*UPL + h
                    a = r*RB + (((unit ^ sw(r)) & (UR-1)) << 4)
                    slots.setdefault((a % 256)//16, set()).add(a)
                worst = max(worst, max(len(v) for v in slots.values()))
    return worst
def lin(mat, nb):   # mat: list of nb 4-bit masks
    def f(r):
        v = 0
        for b, m in enumerate(mat):
            v |= (bin(r & m).count("1") & 1) << b
        return v
    return f
for RB, UPL in ((128,1),(128,2),(64,1)):
    nb = 3 if RB == 128 else 2
    best = None
    for mat in itertools.product(range(16), repeat=nb):
        w = worst_read(RB, UPL, lin(mat, nb))
        if best is None or w < best[0]:
            best = (w, mat)
            if w == 1: break
    print(RB, UPL, best)
print("verify:")
print(worst_read(128,1,lambda r:(r>>1)&7), worst_read(128,2,lambda r:((r>>1)&1)|(r&4)), worst_read(64,1,lambda r:((r>>2)&1)<<1))
