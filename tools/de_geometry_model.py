# python model of de.hip's compile-time geometry + LDS bank-conflict count of the tap loop's ds_read_b128
def rne_half(v):
    if v % 2 == 0: return v // 2
    a = (v - 1) // 2
    return a if a % 2 == 0 else a + 1
NX = [2, 0, 2, -2, 2, -1, 2, 1]; NY = [0, 2, 2, 2, 1, 2, -1, 2]; K = [0, 0, 2, -2, 4, -1, -4, 1]
dx = lambda P, r: rne_half(NX[P] * r); dy = lambda P, r: rne_half(NY[P] * r)
def fh(v): return v // 2 if v >= 0 else -((-v + 1) // 2)
shear = lambda P, j: fh(j * K[P])
dv = lambda P, par, x, y: x - (shear(P, par + y) - shear(P, par))
hoisted = lambda P: P < 4
def reach(P, blur):
    hu = hv = 0
    for par in range(2):
        for r in range(-16, 17):
            for i in range(-3, 4):
                for j in range(-3, 4):
                    if not blur and (i or j or abs(r) == 16): continue
                    if abs(r) == 16 and (i or j or hoisted(P)): continue
                    u = dy(P, r); v = dv(P, par, dx(P, r), dy(P, r))
                    p1 = (par + u) & 1
                    v += dv(P, p1, dx(P, 2 * i), dy(P, 2 * i)); u += dy(P, 2 * i)
                    p2 = (par + u) & 1
                    v += dv(P, p2, dx(P, j), dy(P, j)); u += dy(P, j)
                    hu = max(hu, abs(u)); hv = max(hv, abs(v))
    return hu, hv
def geo(P, TW, TH):
    HU, HV = reach(P, True); HBU, HBV = reach(P, False); HA = abs(dy(P, 16))
    g = dict(P=P, TW=TW, TH=TH, HU=HU, HV=HV, HA=HA, HBU=HBU, HBV=HBV, ROWS=TH + 2 * HU, COLS=TW + 2 * HV,
             AROWS=TH + 2 * HA, BROWS=TH + 2 * HBU, BCOLS=TW + 2 * HBV)
    g['NPX'] = g['ROWS'] * g['COLS']; g['NPXA'] = g['AROWS'] * g['COLS']; g['NPXB'] = g['BROWS'] * g['BCOLS']
    g['LDS'] = (g['NPXA'] + g['NPXB']) * 16 + (g['NPX'] * 4 if hoisted(P) else 0) + 64
    return g
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]
def out_px(P, TW, wv, lane):
    if P == 0: WPR = TW // 64; return wv // WPR, (wv % WPR) * 64 + lane
    if K[P] & 1:
        RPW = 64 // TW
        return (wv >> 1) * 2 * RPW + (wv & 1) + 2 * (lane // TW), lane % TW
    return wv * (64 // TW) + lane // TW, lane % TW
def b128_cycles(elems):
    # elems: element index (float4 units) per lane; cycles = sum over groups of max multiplicity on any bank
    tot = 0
    for g in GROUPS:
        cnt = {}
        for l in g:
            b = elems[l] % 16          # 16-byte slot within the 256-byte bank row
            cnt.setdefault(b, set()).add(elems[l])
        tot += max(len(s) for s in cnt.values())
    return tot
if __name__ == '__main__':
    shapes = {0: (64, 8), 1: (8, 32), 2: (8, 32), 3: (8, 32), 4: (16, 32), 5: (16, 32), 6: (16, 32), 7: (16, 32)}
    for P in range(8):
        TW, TH = shapes[P]; g = geo(P, TW, TH)
        ea = [(out_px(P, TW, 0, l)[0] + g['HA']) * g['COLS'] + out_px(P, TW, 0, l)[1] + g['HV'] for l in range(64)]
        eb = [(out_px(P, TW, 0, l)[0] + g['HBU']) * g['BCOLS'] + out_px(P, TW, 0, l)[1] + g['HBV'] for l in range(64)]
        print(g, 'A b128 cycles', b128_cycles(ea), 'B b128 cycles', b128_cycles(eb), 'staged/out %.2f' % (g['NPX'] / (TW * TH)))
