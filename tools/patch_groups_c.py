def rep(path,a,b,count=1):
    s=open(path).read()
    assert s.count(a)==count,(path,s.count(a),a)
    open(path,'w').write(s.replace(a,b))
H='/root/repo/include/stlpose_hip.h'
rep(H,'''#define STL_OP_HEAD_BWD 7
''','''#define STL_OP_HEAD_BWD 7
#define STL_OP_REDUCE_RANGE 8 /* stl_reduce_slabs over a sub-range of the table (a gradient bucket) */
#define STL_OP_BN_GRADS_RANGE 9 /* stl_bn_param_grads over a sub-range of the table */
/* A gradient bucket = a contiguous slice of the flat gradient buffer whose weight-gradient slabs and
 * BatchNorm reductions are complete at some point of the backward program.  Reducing it there (and
 * recording an event) lets the data-parallel all-reduce of that slice start while the rest of
 * backward still runs (reference: the per-step gradient gather of nn.DataParallel, 02_train.py:109). */
typedef struct stl_reduce_range { const float* partials; float* grads; const stl_slab* tab; int32_t n, blk_base, nblocks, pad_; } stl_reduce_range;
typedef struct stl_bn_range { const double* rstats; float* grads; const stl_bnrec* tab; int32_t n, pad_; } stl_bn_range;
''')
rep(H,'''    int32_t nwait;
    int32_t wait[6];''','''    int32_t nwait;
    int32_t wait[8];''')
rep(H,'''int stl_program_destroy(void* program);
''','''int stl_program_destroy(void* program);
/* Make `stream` wait for op `op` (which must record) of the LAST run of the program: how a
 * communication stream picks up a finished gradient bucket. */
int stl_program_wait_op(void* program, int op, void* stream);
''')
rep(H,'''int stl_reduce_slabs(const float* partials, float* grads, const stl_slab* tab, int n, int nblocks,''','''int stl_reduce_slabs_range(const stl_reduce_range* r, void* stream);
int stl_reduce_slabs(const float* partials, float* grads, const stl_slab* tab, int n, int nblocks,''')

E='/root/repo/stlpose_amd/csrc/elementwise.hip'
rep(E,'''__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* partials, float* grads, const stl_slab* tab, int n) {
    const int ei = find_entry(&tab[0].blk0, sizeof(stl_slab) / 4, n, blockIdx.x);''','''__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* partials, float* grads, const stl_slab* tab, int n, int blk_base) {
    const int bx = blockIdx.x + blk_base;  // tab points at the first entry of the range, blk0 values are table-absolute
    const int ei = find_entry(&tab[0].blk0, sizeof(stl_slab) / 4, n, bx);''')
s=open(E).read()
i0=s.index('__global__ __launch_bounds__(256) void reduce_slabs_kernel')
i1=s.index('__global__ __launch_bounds__(256) void bn_running_kernel')
seg=s[i0:i1]
assert seg.count('(blockIdx.x - e.blk0)')==1
seg=seg.replace('(blockIdx.x - e.blk0)','(bx - e.blk0)')
s=s[:i0]+seg+s[i1:]
open(E,'w').write(s)
rep(E,'''    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(nblocks), dim3(256), 0, ST, partials, grads, tab, n);
    STL_LAUNCH_CHECK("reduce_slabs");
    return 0;
}''','''    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(nblocks), dim3(256), 0, ST, partials, grads, tab, n, 0);
    STL_LAUNCH_CHECK("reduce_slabs");
    return 0;
}

extern "C" int stl_reduce_slabs_range(const stl_reduce_range* r, void* stream) {
    STL_CHECK(r && r->partials && r->grads && r->tab && r->n >= 0 && r->nblocks >= 0, "reduce_slabs_range: bad arguments");
    if (r->n == 0 || r->nblocks == 0) return 0;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(r->nblocks), dim3(256), 0, ST, r->partials, r->grads, r->tab, r->n, r->blk_base);
    STL_LAUNCH_CHECK("reduce_slabs_range");
    return 0;
}''')

P='/root/repo/stlpose_amd/csrc/program.hip'
rep(P,'''o.nwait > 6''','''o.nwait > 8''')
rep(P,'''            default: return stl_set_error("program_run: op %d has unknown kind %d", i, o.kind);''','''            case STL_OP_REDUCE_RANGE: rc = stl_reduce_slabs_range(static_cast<const stl_reduce_range*>(o.desc), st); break;
            case STL_OP_BN_GRADS_RANGE: {
                const stl_bn_range* a = static_cast<const stl_bn_range*>(o.desc);
                rc = stl_bn_param_grads(a->rstats, a->grads, a->tab, a->n, st);
                break;
            }
            default: return stl_set_error("program_run: op %d has unknown kind %d", i, o.kind);''')
rep(P,'''extern "C" int stl_program_run(''','''extern "C" int stl_program_wait_op(void* h, int op, void* stream) {
    Program* p = static_cast<Program*>(h);
    STL_CHECK(p && op >= 0 && op < (int)p->ops.size() && p->ev_of[op] >= 0, "program_wait_op: op %d does not record an event", op);
    STL_CHECK(hipStreamWaitEvent((hipStream_t)stream, p->ev[p->ev_of[op]], 0) == hipSuccess, "program_wait_op: wait failed");
    return 0;
}

extern "C" int stl_program_run(''')

Cp='/root/repo/stlpose_amd/capi.py'
rep(Cp,'''("nwait", i32), ("wait", i32 * 6), ("record", i32)]''','''("nwait", i32), ("wait", i32 * 8), ("record", i32)]


class ReduceRange(C.Structure):
    _fields_ = [("partials", vp), ("grads", vp), ("tab", vp), ("n", i32), ("blk_base", i32), ("nblocks", i32), ("pad_", i32)]


class BNRange(C.Structure):
    _fields_ = [("rstats", vp), ("grads", vp), ("tab", vp), ("n", i32), ("pad_", i32)]''')
rep(Cp,'''"stl_upsample_backward": 4, "stl_patch3x3": 5, "stl_head_forward": 6, "stl_head_backward": 7}''','''"stl_upsample_backward": 4, "stl_patch3x3": 5, "stl_head_forward": 6, "stl_head_backward": 7,
           "stl_reduce_slabs_range": 8, "stl_bn_grads_range": 9}''')
rep(Cp,'''    "stl_reduce_slabs": [vp, vp, vp, i32, i32, vp],''','''    "stl_reduce_slabs": [vp, vp, vp, i32, i32, vp],
    "stl_reduce_slabs_range": [C.POINTER(ReduceRange), vp],''')
rep(Cp,'''    "stl_program_destroy": [vp],''','''    "stl_program_destroy": [vp],
    "stl_program_wait_op": [vp, i32, vp],''')
