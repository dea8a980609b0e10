import itertools
B128_GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
B128_GROUPS += [[l+32 for l in g] for g in B128_GROUPS]
def cost(addrs_dw, width, nbanks):
    """addrs: list of (first dword) per lane in one group -> LDS cycles = max distinct addresses per bank"""
    per_bank = {}
    for a in addrs_dw:
        for k in range(width):
            per_bank.setdefault((a + k) % nbanks, set()).add(a + k)
    return max(len(v) for v in per_bank.values())
def write_b64(addr_of_lane):      # 4 groups of 16 contiguous lanes, 32 banks
    return sum(cost([addr_of_lane(l) for l in range(g*16, g*16+16)], 2, 32) for g in range(4))
def read_b128(addr_of_lane):
    return sum(cost([addr_of_lane(l) for l in grp], 4, 64) for grp in B128_GROUPS)
def read_b64(addr_of_lane):
    return sum(cost([addr_of_lane(l) for l in range(h*32, h*32+32)], 2, 64) for h in range(2))

def analyse(CO_B, CSZ, CSX, swz):
    QZ = CO_B // 4
    # --- Z writes: wave w (tid = w*64 + lane), unit u = tid
    zw = 0; n = 0
    for w in range(4):
        for e in range(4):
            def a(l, w=w, e=e):
                u = w*64 + l; pg, q = divmod(u, QZ); r, cg = pg >> 2, pg & 3
                ch = 4*q + e
                return ch*CSZ + ((r*8 + cg*2) ^ swz(ch))
            zw += write_b64(a); n += 4
    # --- Z reads
    zr = 0; m = 0
    for s in range(4):
        def a(l, s=s):
            li, g = l & 15, l >> 4; rr, h = 2*s + (g >> 1), g & 1
            return li*CSZ + ((rr*8 + h*4) ^ swz(li))
        zr += read_b128(a); m += 4
    # --- X writes (QA = 8 for 32 ch): units 50*8 = 400 -> waves: u = tid + i*256
    QA = CO_B // 4
    xw = 0; nx = 0
    for i in range(2):
        for w in range(4):
            for e in range(4):
                lanes = [w*64 + l + i*256 for l in range(64)]
                def a(l, w=w, e=e, i=i):
                    u = w*64 + l + i*256
                    if u >= 50*QA: return None
                    pg, q = divmod(u, QA); hr, hg = divmod(pg, 5); ch = 4*q + e
                    return ch*CSX + ((hr*12 + hg*2) ^ swz(ch))
                for g in range(4):
                    ad = [a(l) for l in range(g*16, g*16+16)]
                    ad = [x for x in ad if x is not None]
                    if ad: xw += cost(ad, 2, 32); nx += 1
    xr = 0; mx = 0; xr64 = 0
    for s in range(4):
        for dy in range(3):
            def a(l, s=s, dy=dy):
                li, g = l & 15, l >> 4; rr, h = 2*s + (g >> 1), g & 1
                return li*CSX + (((rr+dy)*12 + h*4) ^ swz(li))
            def a2(l, s=s, dy=dy):
                li, g = l & 15, l >> 4; rr, h = 2*s + (g >> 1), g & 1
                return li*CSX + (((rr+dy)*12 + h*4 + 4) ^ swz(li))
            xr += read_b128(a); mx += 4; xr64 += read_b64(a2)
    return zw / n, zr / m, xw / nx, xr / mx, xr64 / (12*2)

print("current:", analyse(32, 68, 124, lambda ch: 0))
for csz, csx in ((72,120),(72,136),(68,136)):
    print(csz, csx, analyse(32, csz, csx, lambda ch: 0))

fams = {
 "q": lambda ch: ((ch >> 2) & 7) << 2,
 "q3": lambda ch: ((ch >> 2) & 3) << 2,
 "c7": lambda ch: (ch & 7) << 2,
 "c3": lambda ch: (ch & 3) << 2,
 "h": lambda ch: ((ch >> 1) & 7) << 2,
 "qx": lambda ch: (((ch >> 2) ^ ch) & 7) << 2,
 "q_hi": lambda ch: ((ch >> 2) & 7) << 3,
 "q1": lambda ch: ((ch >> 2) & 1) << 2,
 "q2b": lambda ch: ((ch >> 2) & 3) << 3,
 "none": lambda ch: 0,
}
best = []
for name, f in fams.items():
    for csz in range(64, 100, 4):
        for csx in range(120, 160, 4):
            r = analyse(32, csz, csx, f)
            best.append((sum(r) , r, name, csz, csx))
best.sort(key=lambda t: t[0])
for b in best[:15]: print(b)
print("---- X only")
import random
def xcost(csx, f):
    r = analyse(32, 72, csx, f)
    return r[2], r[3], r[4]
cands = []
perms = []
# general: swz = 4 * T[ch & 15] ^ 4 * U[(ch>>4)&1]?  search table T over 16 channels within a 16-row sub-tile (reads see li only), values 0..7
# writes see ch = 4q+e for q = 0..7 (lanes) -> table over 32 channels; reads use li = ch & 15 (wj*16 + li: swz must depend on ch incl. wj)
random.seed(1)
best = (9, None)
for csx in (128, 132, 136, 140, 144, 148, 152, 156, 160):
    for name, f in fams.items():
        c = xcost(csx, f)
        cands.append((c[0] + 3*c[1] + 1.5*c[2], c, name, csx))
cands.sort(key=lambda t: t[0])
for c in cands[:10]: print(c)
print("---- X small strides")
fams2 = dict(fams)
for a_ in range(1, 8):
    for sh in (0, 1, 2):
        fams2[f"m{a_}s{sh}"] = (lambda ch, a_=a_, sh=sh: (((ch >> sh) * a_) & 7) << 2)
        fams2[f"x{a_}s{sh}"] = (lambda ch, a_=a_, sh=sh: ((((ch >> sh) * a_) ^ (ch >> 3)) & 7) << 2)
out = []
for csx in (128, 132, 136):
    for name, f in fams2.items():
        c = xcost(csx, f)
        out.append((c[0] + 3*c[1] + 1.5*c[2], c, name, csx))
out.sort(key=lambda t: t[0])
for c in out[:12]: print(c)
print("---- Z with these")
for name in set(n for _, _, n, _ in out[:12]):
    for csz in (64, 68, 72, 76, 80):
        r = analyse(32, csz, 136, fams2[name]); print(name, csz, r[:2])
print("---- 16-channel blocks")
for name in ("none", "q", "qx", "m1s2"):
    for csz, csx in ((68, 124), (72, 136), (72, 132), (68, 136)):
        try:
            print(name, csz, csx, analyse(16, csz, csx, fams2[name]))
        except Exception as e:
            print(name, csz, csx, "err", e)

print("---- X second read as b128 / alternatives")
def xreads(csx, f, second):
    tot = 0; n = 0
    for s in range(4):
        for dy in range(3):
            def a(l, s=s, dy=dy):
                li, g = l & 15, l >> 4; rr, h = 2*s + (g >> 1), g & 1
                return li*csx + (((rr+dy)*12 + h*4) ^ f(li))
            def a2(l, s=s, dy=dy):
                li, g = l & 15, l >> 4; rr, h = 2*s + (g >> 1), g & 1
                return li*csx + (((rr+dy)*12 + h*4 + 4) ^ f(li))
            tot += read_b128(a) + (read_b128(a2) if second == "b128" else 2 * read_b64(a2)); n += 1
    return tot / n      # LDS cycles per (s, dy): b128 = 4 groups, b64 = 2 halves
res = []
for csx in (128, 132, 136):
    for name, f in fams2.items():
        for second in ("b64", "b128"):
            w = analyse(32, 72, csx, f)[2]
            res.append((xreads(csx, f, second), w, name, csx, second))
res.sort()
for r in res[:12]: print(r)
