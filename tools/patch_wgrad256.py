p='/root/repo/stlpose_amd/csrc/wgrad.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)>=1, a
    s=s.replace(a,b) if cnt==0 else s.replace(a,b,cnt)
rep('template <typename T, int KS, int NVH, bool GQ>\n__global__','template <typename T, int KS, int NVH, bool GQ, int TPX>\n__global__')
rep('constexpr int NVG = 128 * VPX / 256;','constexpr int NVG = TPX * VPX / 256;')
rep('constexpr int NKS = (128 / KSTEP + 3) / 4;','constexpr int NKS = (TPX / KSTEP + 3) / 4;')
rep('if (m > 127) m = 127;','if (m > TPX - 1) m = TPX - 1;')
rep('''template <typename T, int KS, int NVH, bool GQ>
int launch(''','''template <typename T, int KS, int NVH, bool GQ, int TPX = 128>
int launch(''')
rep('wgrad_kernel<T, KS, NVH, GQ>),\n','wgrad_kernel<T, KS, NVH, GQ, TPX>),\n')
rep('hipLaunchKernelGGL((wgrad_kernel<T, KS, NVH, GQ>)','hipLaunchKernelGGL((wgrad_kernel<T, KS, NVH, GQ, TPX>)')
rep('''    const bool gq = k.p.g.mode == STL_SRC_BNBWD;
''','''    const bool gq = k.p.g.mode == STL_SRC_BNBWD;
    if (k.p.TH * k.p.TW > 128) {  // 256-pixel tiles: bf16, halo of at most 6 vectors per thread
        if constexpr (sizeof(T) == 2) {
            if (nvh <= 6) return gq ? launch<T, KS, 6, true, 256>(k, grid, lds, st) : launch<T, KS, 6, false, 256>(k, grid, lds, st);
        }
        return stl_set_error("wgrad: 256-pixel tiles need bf16 and a halo of at most 384 pixels (have %d)", k.HP);
    }
''')
rep('STL_CHECK(p.TH >= 1 && p.TW >= 1 && p.TH * p.TW <= 128, "wgrad: tile exceeds 128 pixels");','STL_CHECK(p.TH >= 1 && p.TW >= 1 && p.TH * p.TW <= 256, "wgrad: tile exceeds 256 pixels");')
rep('int szG = 128 * k.psg;','int szG = (p.TH * p.TW > 128 ? 256 : 128) * k.psg;')
rep('// at most 2 K steps per wave (128 px / KSTEP / 4 waves: 1 for bf16, 2 for fp32)','// K steps per wave: TPX px / KSTEP / 4 waves (128 px: 1 for bf16, 2 for fp32; 256 px bf16: 2)')
open(p,'w').write(s)

p='/root/repo/stlpose_amd/engine.py'
s=open(p).read()
rep('def choose_tile(B: int, Ho: int, Wo: int, stride: int, ks: int, esz: int, bn_cols: int = 64) -> Tuple[int, int]:','def choose_tile(B: int, Ho: int, Wo: int, stride: int, ks: int, esz: int, bn_cols: int = 64,\n                maxpx: int = 128, maxhalo: int = 576) -> Tuple[int, int]:')
rep('''    best, best_tile = -1.0, (1, min(Wo, 128))
    for tw in range(min(Wo, 4), min(Wo, 128) + 1):
        th = max(1, min(128 // tw, vrows))''','''    best, best_tile = -1.0, (1, min(Wo, maxpx))
    for tw in range(min(Wo, 4), min(Wo, maxpx) + 1):
        th = max(1, min(maxpx // tw, vrows))''')
rep('if lds > 150 * 1024 or hr * hc > 576:','if lds > 150 * 1024 or hr * hc > maxhalo:')
rep('eff = (Wo / cols) * (B * Ho / rows) * (th * tw / 128.0)','eff = (Wo / cols) * (B * Ho / rows) * (th * tw / float(maxpx))')
rep('''                wg.TH, wg.TW = choose_tile(x.B, y.H, y.W, kstride, kks, self.esz, bn_cols=32)
''','''                big = (self.esz == 2 and kstride == 1 and os.environ.get("STLPOSE_WGRAD_TILE", "256") == "256"
                       and x.B * y.H * y.W >= 256 * 64)
                if big:   # 256-pixel tiles halve the per-pixel barrier / latency cost of the K loop
                    wg.TH, wg.TW = choose_tile(x.B, y.H, y.W, kstride, kks, self.esz, bn_cols=32, maxpx=256, maxhalo=384)
                else:
                    wg.TH, wg.TW = choose_tile(x.B, y.H, y.W, kstride, kks, self.esz, bn_cols=32)
''')
open(p,'w').write(s)
