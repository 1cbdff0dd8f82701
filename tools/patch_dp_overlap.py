def rep(path,a,b,count=1):
    s=open(path).read()
    assert s.count(a)==count,(path,s.count(a),a)
    open(path,'w').write(s.replace(a,b))
D='/root/repo/stlpose_amd/dp.py'
rep(D,'''    def __init__(self, flat: torch.Tensor, process_group=None, bucket_mb: float = 32.0):
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        n = flat.numel()
        nb = max(1, int(round(n * flat.element_size() / (bucket_mb * 2 ** 20))))
        edges = [int(round(i * n / nb)) for i in range(nb + 1)]
        # reversed: the tail of the buffer (last layers) is ready first in backward
        self.bounds = [(edges[i], edges[i + 1]) for i in reversed(range(nb))]''','''    def __init__(self, flat: torch.Tensor, process_group=None, bucket_mb: float = 32.0, bounds=None):
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        n = flat.numel()
        if bounds is not None:
            # the engine's gradient buckets, in the order backward completes them (engine.buckets)
            self.bounds = [(int(a), int(b)) for a, b in bounds]
            assert sorted(self.bounds) == sorted(set(self.bounds)) and sum(b - a for a, b in self.bounds) == n
        else:
            nb = max(1, int(round(n * flat.element_size() / (bucket_mb * 2 ** 20))))
            edges = [int(round(i * n / nb)) for i in range(nb + 1)]
            # reversed: the tail of the buffer (last layers) is ready first in backward
            self.bounds = [(edges[i], edges[i + 1]) for i in reversed(range(nb))]''')
rep(D,'''    def launch(self, upto: Optional[int] = None):
        """Start async all-reduces for buckets [len(started), upto)."""
        if self.world == 1:
            return''','''    def launch(self, upto: Optional[int] = None, force: bool = False):
        """Start async all-reduces for buckets [len(started), upto)."""
        if self.world == 1 and not force:
            return''')
T='/root/repo/stlpose_amd/train_step.py'
rep(T,'''        self.dp = FlatAllReduce(self.store.grads, process_group, bucket_mb) if process_group is not None else None''','''        # buckets = the engine's own gradient buckets (contiguous slices, final at known points of backward)
        self.dp = (FlatAllReduce(self.store.grads, process_group, bucket_mb, bounds=[(b["lo"], b["hi"]) for b in self.eng.buckets])
                   if process_group is not None else None)
        self._comm: Optional[torch.cuda.Stream] = None
        self._force_dp = os.environ.get("STLPOSE_DP_FORCE", "0") == "1"   # exercise the DP path with one rank (tests)''')
rep(T,'''    def _allreduce(self):
        if self.dp is not None and self.world > 1:
            self.dp.all_reduce()''','''    def _allreduce(self):
        """Bucketed all-reduce overlapped with backward: the backward program is already ENQUEUED when
        this runs; every bucket's collective is issued on a communication stream that waits for the
        event recorded after that bucket's slab reduction, so RCCL works on the last layers'
        gradients while the data-gradient chain is still in the early layers.  The main stream then
        waits for all collectives before the optimiser."""
        if self.dp is None or (self.world == 1 and not self._force_dp):
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=self.dev)
        for i in range(len(self.eng.buckets)):
            self.eng.bucket_wait(i, self._comm.cuda_stream)
            with torch.cuda.stream(self._comm):
                self.dp.launch(upto=i + 1, force=self._force_dp)
        self.dp.wait()''')
