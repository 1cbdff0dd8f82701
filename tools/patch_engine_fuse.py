p='/root/repo/stlpose_amd/engine.py'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''    dt: Optional[torch.Tensor] = None                        # bn: grad wrt BN output (masked)
    consumers: int = 0
''','''    dt: Optional[torch.Tensor] = None                        # bn: grad wrt BN output (masked)
    consumers: int = 0
    bwd_seen: int = 0                                        # consumers already handled by the backward builder
    fused_du: Optional[torch.Tensor] = None                  # plain: masked gradient produced by a fused dgrad
''')
rep('''        for node in reversed(self.tape):
            kind = node[0]
            if kind == "head":
                _, x, key, joints = node
                nblk = 256''','''        producer = {id(n[2]): n for n in self.tape if n[0] == "fuse"}
        fuse_block_end = os.environ.get("STLPOSE_FUSE_BLOCK_END", "1") != "0"
        for node in reversed(self.tape):
            kind = node[0]
            if kind == "head":
                _, x, key, joints = node
                x.bwd_seen += 1
                nblk = 256''')
# fuse node
rep('''                _, terms, z, relu, strm = node
                assert 1 <= len(z.grads) <= 4, f"fuse output has {len(z.grads)} gradient contributions"
                p = capi.FuseBwd()''','''                _, terms, z, relu, strm = node
                for a, _s, _ in terms:
                    a.bwd_seen += 1
                if z.fused_du is not None:
                    # the ReLU mask, the BatchNorm reductions and the sum of contributions were done in the
                    # epilogue of the data gradient that produced the last contribution (mask_z)
                    assert not z.grads
                    z.grads.append(z.fused_du)
                assert 1 <= len(z.grads) <= 4, f"fuse output has {len(z.grads)} gradient contributions"
                p = capi.FuseBwd()''')
rep('''                trivial = (len(z.grads) == 1 and not relu and not same_bn)''','''                trivial = (len(z.grads) == 1 and not relu and not same_bn) or z.fused_du is not None''')
# conv node
rep('''                _, x, y, ci, (kks, kstride), strm = node
                assert y.consumers == 1''','''                _, x, y, ci, (kks, kstride), strm = node
                x.bwd_seen += 1
                assert y.consumers == 1''')
rep('''                if x.kind == "plain":
                    out = self._new_grad(x)
                    if x.grads:
                        ad = x.grads.pop()
                        d.addend = ad.data_ptr()
                        dreads.append(ad.data_ptr())
                    x.grads.append(out)
                else:''','''                if x.kind == "plain":
                    out = self._new_grad(x)
                    if x.grads:
                        ad = x.grads.pop()
                        d.addend = ad.data_ptr()
                        dreads.append(ad.data_ptr())
                    # Residual block end z = ReLU(BN(y) + skip): when this data gradient is the LAST
                    # contribution to dz, its epilogue also applies the ReLU mask and reduces the
                    # BatchNorm-backward sums, so no separate pass over dz / z / y is needed.
                    F = producer.get(id(x))
                    same_bn = [a for a, s_, _ in F[1] if a.kind == "bn" and s_ == 0] if F else []
                    if (fuse_block_end and F is not None and F[3] and len(same_bn) == 1 and not x.grads
                            and x.bwd_seen == x.consumers):
                        ybn = same_bn[0]
                        d.mask_z = x.ptr
                        d.mask_y = ybn.ptr
                        d.mask_bn = self._src(ybn, relu=False)
                        d.red = self.rstats.data_ptr() + 8 * ybn.bn.stats_off
                        dreads += [x.ptr, ybn.ptr]
                        x.fused_du = out
                    else:
                        x.grads.append(out)
                else:''')
open(p,'w').write(s)
