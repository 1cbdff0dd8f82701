// Native replay of a planned kernel program: one C call enqueues every launch of the forward or
// backward plan on its HIP streams, with the cross-stream dependencies expressed as events.
// (The Python planner builds the op list once; per step the host does O(1) Python work.)
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.cuh"

namespace {
struct Program {
    std::vector<stl_op> ops;
    std::vector<hipEvent_t> ev;  // one per op with record != 0
    std::vector<int> ev_of;      // op index -> event index or -1
    hipEvent_t fork = nullptr;
    std::vector<hipEvent_t> join;
    int nstreams = 1;
    // explicit HIP graph of the program (stl_program_graph_build): one kernel node per recorded launch
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<StlLaunchRec> recs;   // own the argument copies the kernel nodes point at
    int nnodes = 0;
};
}  // namespace

// Flags of the program's cross-stream events.  A default HIP event performs a SYSTEM-scope fence (cache write-back and
// invalidation, "and the performance impact of those actions on the execution of following work", hip_runtime_api.h) when it is
// recorded; producers and consumers of a program's events are kernels on the SAME device, for which device scope orders the data
// (the host reads results only behind a stream / device synchronisation, which fences on its own).  Round 3, one call, two rounds:
// 14.69-14.71 ms per step with the default flags, 14.52-14.57 with hipEventDisableSystemFence, 14.66-14.71 with
// hipEventReleaseToDevice; stream memory operations (hipStreamWriteValue32 / WaitValue32) instead of events: 15.45 (removed).
// Round 5: STLPOSE_EVENT_FENCE=system (read when a program is created) restores the default, fencing events -- the A/B arm of
// tests/test_visibility_gpu.py, which hands 64 MB buffers from stream to stream through a program 2000 times and replays the
// benchmarked train step 300 times under both settings; no stale read under either (the round-4 wrong result was
// an undersized statistics arena, engine.bn_weight_keys, not visibility: every kernel dispatch carries its own agent-scope
// acquire and release, the event's fence only adds the system scope).
static unsigned event_flags() {
    const char* e = getenv("STLPOSE_EVENT_FENCE");
    if (e && strcmp(e, "system") == 0) return hipEventDisableTiming;
    return hipEventDisableTiming | hipEventDisableSystemFence;
}

// one op of a program on stream `st` (also what the graph builder calls, with the launch recorder set)
static int run_op(const stl_op& o, void* st, int i) {
    int rc;
        switch (o.kind) {
            case STL_OP_CONV: rc = stl_conv_forward(static_cast<const stl_conv*>(o.desc), st); break;
            case STL_OP_WGRAD: rc = stl_conv_wgrad(static_cast<const stl_wgrad*>(o.desc), st); break;
            case STL_OP_WGRAD_GROUP: rc = stl_conv_wgrad_group(static_cast<const stl_wgrad_group*>(o.desc), st); break;
            case STL_OP_FUSE: rc = stl_fuse_forward(static_cast<const stl_fuse*>(o.desc), st); break;
            case STL_OP_FUSE_BWD: rc = stl_fuse_backward(static_cast<const stl_fuse_bwd*>(o.desc), st); break;
            case STL_OP_UP_BWD: rc = stl_upsample_backward(static_cast<const stl_upbwd*>(o.desc), st); break;
            case STL_OP_PATCH: {
                const stl_patch* a = static_cast<const stl_patch*>(o.desc);
                rc = stl_patch3x3(a->dtype, a->img, a->out, a->B, a->H, a->W, a->stride, a->mean3, a->std3, st);
                break;
            }
            case STL_OP_HEAD: {
                const stl_head* a = static_cast<const stl_head*>(o.desc);
                rc = stl_head_forward(a->dtype, a->x, a->w, a->bias, a->out, a->B, a->H, a->W, a->Ci, a->J, st);
                break;
            }
            case STL_OP_HEAD_BWD: {
                const stl_head_bwd* a = static_cast<const stl_head_bwd*>(o.desc);
                rc = stl_head_backward(a->dtype, a->x, a->w, a->dout, a->dx, a->partial, a->nblk, a->B, a->H, a->W, a->Ci, a->J, st);
                break;
            }
            case STL_OP_OPTIM_SLICE: {
                const stl_optim_slice* a = static_cast<const stl_optim_slice*>(o.desc);
                rc = a->kind == 0 ? stl_adam_slice(a->p, a->g, a->m, a->v, a->n, a->hyper, a->step, st)
                                  : stl_sgd_slice(a->p, a->g, a->m, a->n, a->hyper, a->step, st);
                break;
            }
            case STL_OP_WPREP_RANGE: {
                const stl_wprep_range* a = static_cast<const stl_wprep_range*>(o.desc);
                rc = stl_weight_prep_range(a->dtype, a->master, a->wk, a->tab, a->n, a->blk_base, a->nblocks, st);
                break;
            }
            case STL_OP_REDUCE_RANGE: rc = stl_reduce_slabs_range(static_cast<const stl_reduce_range*>(o.desc), st); break;
            case STL_OP_BN_GRADS_RANGE: {
                const stl_bn_range* a = static_cast<const stl_bn_range*>(o.desc);
                rc = stl_bn_param_grads(a->rstats, a->grads, a->tab, a->n, st);
                break;
            }
            default: return stl_set_error("program_run: op %d has unknown kind %d", i, o.kind);
        }
    return rc;
}

extern "C" int stl_program_create(const stl_op* ops, int n, int nstreams, void** out) {
    STL_CHECK(ops && out && n >= 0 && nstreams >= 1 && nstreams <= 16, "program_create: bad arguments");
    Program* p = new Program();
    p->ops.assign(ops, ops + n);
    p->nstreams = nstreams;
    p->ev_of.assign(n, -1);
    for (int i = 0; i < n; ++i) {
        const stl_op& o = ops[i];
        if (o.stream < 0 || o.stream >= nstreams || o.nwait < 0 || o.nwait > 8 || !o.desc) {
            delete p;
            return stl_set_error("program_create: op %d malformed (stream %d, nwait %d)", i, o.stream, o.nwait);
        }
        for (int w = 0; w < o.nwait; ++w)
            if (o.wait[w] < 0 || o.wait[w] >= i || !ops[o.wait[w]].record) {
                delete p;
                return stl_set_error("program_create: op %d waits on op %d which does not record an earlier event", i, o.wait[w]);
            }
        if (o.record) {
            hipEvent_t e;
            // gradient-bucket ends are picked up from OUTSIDE the program (stl_program_wait_op: the communication stream of a
            // data-parallel run, whose collective may hand the bucket to other devices): those few keep the system-scope fence
            if (hipEventCreateWithFlags(&e, o.kind == STL_OP_BN_GRADS_RANGE ? (unsigned)hipEventDisableTiming : event_flags()) != hipSuccess) {
                delete p;
                return stl_set_error("program_create: hipEventCreate failed");
            }
            p->ev_of[i] = (int)p->ev.size();
            p->ev.push_back(e);
        }
    }
    (void)hipEventCreateWithFlags(&p->fork, event_flags());
    p->join.resize(nstreams);
    for (int s = 0; s < nstreams; ++s) (void)hipEventCreateWithFlags(&p->join[s], event_flags());
    *out = p;
    return 0;
}

extern "C" int stl_program_destroy(void* h) {
    Program* p = static_cast<Program*>(h);
    if (!p) return 0;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->join) (void)hipEventDestroy(e);
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    for (StlLaunchRec& r : p->recs) {
        r.del(r.args_owner);
        delete[] r.params;
    }
    if (p->fork) (void)hipEventDestroy(p->fork);
    delete p;
    return 0;
}

extern "C" int stl_program_wait_op(void* h, int op, void* stream) {
    Program* p = static_cast<Program*>(h);
    STL_CHECK(p && op >= 0 && op < (int)p->ops.size() && p->ev_of[op] >= 0, "program_wait_op: op %d does not record an event", op);
    STL_CHECK(hipStreamWaitEvent((hipStream_t)stream, p->ev[p->ev_of[op]], 0) == hipSuccess, "program_wait_op: wait failed");
    return 0;
}

// ops [first, last) of the program; the first range forks the side streams off the main stream, the last one joins them.
// Splitting a run lets the host put other work between two ranges IN ISSUE ORDER: hardware queues are in-order and the
// communication stream of a data-parallel run shares one with a compute stream (GPU_MAX_HW_QUEUES = 4), so a gradient bucket's
// all-reduce must be ENQUEUED right behind the bucket's last op -- issued after the whole backward program it would sit behind
// everything that queue still has to run (measured with a stand-in kernel, profiles/r04_dp_overlap*.txt: bucket 0 final at
// 6.3 ms, its communication-stream kernel started at 13.9 ms).
extern "C" int stl_program_run_range(void* h, void* const* streams, int first, int last) {
    Program* p = static_cast<Program*>(h);
    STL_CHECK(p && streams, "program_run: null program");
    const int n = (int)p->ops.size();
    STL_CHECK(first >= 0 && first <= last && last <= n, "program_run_range: bad range [%d, %d) of %d ops", first, last, n);
    hipStream_t main = (hipStream_t)streams[0];
    if (first == 0 && p->nstreams > 1) {
        STL_CHECK(hipEventRecord(p->fork, main) == hipSuccess, "program_run: fork record failed");
        for (int s = 1; s < p->nstreams; ++s)
            STL_CHECK(hipStreamWaitEvent((hipStream_t)streams[s], p->fork, 0) == hipSuccess, "program_run: fork wait failed");
    }
    for (int i = first; i < last; ++i) {
        const stl_op& o = p->ops[i];
        void* st = streams[o.stream];
        for (int w = 0; w < o.nwait; ++w)
            STL_CHECK(hipStreamWaitEvent((hipStream_t)st, p->ev[p->ev_of[o.wait[w]]], 0) == hipSuccess, "program_run: wait failed");
        int rc = run_op(o, st, i);
        if (rc != 0) return rc;
        if (o.record) STL_CHECK(hipEventRecord(p->ev[p->ev_of[i]], (hipStream_t)st) == hipSuccess, "program_run: record failed");
    }
    if (last == n) {
        for (int s = 1; s < p->nstreams; ++s) {
            STL_CHECK(hipEventRecord(p->join[s], (hipStream_t)streams[s]) == hipSuccess, "program_run: join record failed");
            STL_CHECK(hipStreamWaitEvent(main, p->join[s], 0) == hipSuccess, "program_run: join wait failed");
        }
    }
    return 0;
}

extern "C" int stl_program_run(void* h, void* const* streams) {
    Program* p = static_cast<Program*>(h);
    STL_CHECK(p && streams, "program_run: null program");
    return stl_program_run_range(h, streams, 0, (int)p->ops.size());
}

// ---- explicit HIP graph: the planner already knows every launch and every dependency, so the graph is
// BUILT (hipGraphAddKernelNode + dependency lists), not captured -- stream capture of a plan that forks onto >= 3 streams
// crashes inside hipStreamEndCapture on ROCm 7.2.  Every op is executed once with the launch recorder set (common.cuh,
// STL_LAUNCH): its kernels become nodes, chained in issue order; the first node of an op depends on the last node of the
// previous op of its stream (streams are in-order) and on the last node of every op it waits for.
// a failed build leaves nothing behind: a retry starts from an empty graph and leaks neither nodes nor argument copies
static void graph_discard(Program* p) {
    if (p->graph) (void)hipGraphDestroy(p->graph);
    p->graph = nullptr;
    for (StlLaunchRec& r : p->recs) {
        r.del(r.args_owner);
        delete[] r.params;
    }
    p->recs.clear();
    p->nnodes = 0;
}

extern "C" int stl_program_graph_build(void* h) {
    Program* p = static_cast<Program*>(h);
    STL_CHECK(p, "program_graph_build: null program");
    if (p->exec) return 0;
    const int n = (int)p->ops.size();
    STL_CHECK(hipGraphCreate(&p->graph, 0) == hipSuccess, "program_graph_build: hipGraphCreate failed");
    std::vector<hipGraphNode_t> last_of_op(n, nullptr), last_of_stream(p->nstreams, nullptr);
    for (int i = 0; i < n; ++i) {
        const stl_op& o = p->ops[i];
        StlRecorder rec{nullptr, 0, 0};
        g_stl_recorder = &rec;
        const int rc = run_op(o, nullptr, i);
        g_stl_recorder = nullptr;
        // the program owns every recorded argument copy from here on (graph_discard / stl_program_destroy release them)
        const size_t first_rec = p->recs.size();
        for (int k = 0; k < rec.n; ++k) p->recs.push_back(rec.recs[k]);
        delete[] rec.recs;
        if (rc != 0) {
            graph_discard(p);
            return rc;
        }
        std::vector<hipGraphNode_t> deps;
        if (last_of_stream[o.stream]) deps.push_back(last_of_stream[o.stream]);
        for (int w = 0; w < o.nwait; ++w)
            if (last_of_op[o.wait[w]]) deps.push_back(last_of_op[o.wait[w]]);
        hipGraphNode_t prev = nullptr;
        for (size_t k = first_rec; k < p->recs.size(); ++k) {
            const StlLaunchRec& r = p->recs[k];
            hipKernelNodeParams kp;
            memset(&kp, 0, sizeof(kp));
            kp.func = const_cast<void*>(r.func);
            kp.gridDim = r.grid, kp.blockDim = r.block;
            kp.sharedMemBytes = (unsigned)r.lds;
            kp.kernelParams = r.params;
            kp.extra = nullptr;
            hipGraphNode_t node = nullptr;
            hipError_t e = prev ? hipGraphAddKernelNode(&node, p->graph, &prev, 1, &kp)
                                : hipGraphAddKernelNode(&node, p->graph, deps.empty() ? nullptr : deps.data(), deps.size(), &kp);
            if (e != hipSuccess) {
                graph_discard(p);
                return stl_set_error("program_graph_build: hipGraphAddKernelNode failed for op %d (%s)", i, hipGetErrorString(e));
            }
            prev = node;
            ++p->nnodes;
        }
        if (!prev) {
            // an op without a launch still carries its cross-stream waits: the planner prunes later waits of this stream on the
            // strength of them (engine._schedule, transitive ordering), so they become an empty node that its successors chain to
            if (hipGraphAddEmptyNode(&prev, p->graph, deps.empty() ? nullptr : deps.data(), deps.size()) != hipSuccess) {
                graph_discard(p);
                return stl_set_error("program_graph_build: hipGraphAddEmptyNode failed for op %d", i);
            }
            ++p->nnodes;
        }
        last_of_op[i] = prev, last_of_stream[o.stream] = prev;
    }
    hipError_t e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        graph_discard(p);
        return stl_set_error("program_graph_build: hipGraphInstantiate failed (%s)", hipGetErrorString(e));
    }
    return 0;
}

extern "C" int stl_program_graph_launch(void* h, void* stream) {
    Program* p = static_cast<Program*>(h);
    STL_CHECK(p && p->exec, "program_graph_launch: the program has no graph (stl_program_graph_build)");
    hipError_t e = hipGraphLaunch(p->exec, (hipStream_t)stream);
    STL_CHECK(e == hipSuccess, "program_graph_launch: %s", hipGetErrorString(e));
    return 0;
}
