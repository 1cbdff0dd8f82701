p='/root/repo/stlpose_amd/engine.py'
s=open(p).read()
def rep(a,b,count=1):
    global s
    assert s.count(a)==count,(s.count(a),a)
    s=s.replace(a,b)

# ---- bookkeeping at the start of _build_backward
rep('''        producer = {id(n[2]): n for n in self.tape if n[0] == "fuse"}
        fuse_block_end = os.environ.get("STLPOSE_FUSE_BLOCK_END", "1") != "0"
''','''        producer = {id(n[2]): n for n in self.tape if n[0] == "fuse"}
        fuse_block_end = os.environ.get("STLPOSE_FUSE_BLOCK_END", "1") != "0"
        # ---- gradient buckets: contiguous suffixes of the flat gradient buffer, closed as soon as every
        # parameter in them has its slabs / BatchNorm reductions complete (backward finishes the last
        # layers first).  Each bucket gets one ranged slab reduction + BN-gradient launch inside the
        # program, so the step has no serial tail, and an event a data-parallel all-reduce can wait on.
        self.buckets: List[dict] = []
        bucket_min = int(float(os.environ.get("STLPOSE_BUCKET_MB", "16")) * (1 << 20) / 4)
        bk = dict(done=0, lo=st.nparam, hi=st.nparam, slab0=0, reads=[], strm=0)

        def bucket_add(off: int, size: int):
            bk["done"] += size
            bk["lo"] = min(bk["lo"], off)

        def bucket_close(force: bool = False):
            complete = bk["done"] == bk["hi"] - bk["lo"]          # suffix [lo, hi) fully covered
            if not complete or bk["done"] == 0 or (bk["done"] < bucket_min and not force):
                return
            assert complete
            rr, br = capi.ReduceRange(), capi.BNRange()
            b = dict(lo=bk["lo"], hi=bk["hi"], slab0=bk["slab0"], slab1=len(self.slabs), rr=rr, br=br)
            wstrm = bk["strm"]
            ops.append(("stl_reduce_slabs_range", rr, wstrm, list(bk["reads"]), [("bucket", len(self.buckets))]))
            ops.append(("stl_bn_grads_range", br, wstrm, [("bucket", len(self.buckets))], [("bucketbn", len(self.buckets))]))
            b["op"] = len(ops) - 1
            self.buckets.append(b)
            bk.update(done=0, hi=bk["lo"], slab0=len(self.slabs), reads=[])
''')
# ---- head node: its params + dependency id
rep('''                ops.append(("stl_head_backward", hb, 0, [self.dout.data_ptr(), x.ptr], [dx.data_ptr()]))''','''                ops.append(("stl_head_backward", hb, 0, [self.dout.data_ptr(), x.ptr], [dx.data_ptr(), id(hb)]))
                bk["reads"].append(id(hb))
                bucket_add(st.param_off[key + ".weight"], joints * x.C)
                bucket_add(st.param_off[key + ".bias"], joints)''')
# ---- conv node: after wgrad op appended
rep('''                ops.append(("stl_conv_wgrad", wg, wstrm, [y.dt.data_ptr(), x.ptr], [id(wg)]))''','''                ops.append(("stl_conv_wgrad", wg, wstrm, [y.dt.data_ptr(), x.ptr], [id(wg)]))
                bk["reads"].append(id(wg))
                bk["strm"] = wstrm
                bucket_add(ci.master_off, ci.Co * ci.Ci * ci.ks * ci.ks)
                bucket_add(y.bn.param_off, 2 * y.bn.C)   # gamma, beta of the BatchNorm behind this conv
                conv_done = True''')
rep('''                _, x, y, ci, (kks, kstride), strm = node
                x.bwd_seen += 1''','''                _, x, y, ci, (kks, kstride), strm = node
                conv_done = False
                x.bwd_seen += 1''')
open(p,'w').write(s)
