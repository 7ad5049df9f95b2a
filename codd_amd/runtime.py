"""Per-shape frame plan: the steady-state frame (stereo -> motion -> fusion) captured once into a
hipGraph and replayed per frame.

The reference runs ~3 500-4 000 eager launches per frame; here a frame is
~600 launches of hand-written kernels whose host-side issue cost (ctypes + Python) would still
dominate at 30 fps, so the whole frame is recorded with HIP stream capture
(torch.cuda.CUDAGraph drives hipStreamBeginCapture/EndCapture and owns the capture-time memory
pool).  Recurrent state (reference model/codd.py:322-366: memory, raft_feat, raft_netinp) lives
in static buffers that the graph reads at its head and overwrites at its tail.
"""
import os

import torch

from . import ops


class FrameRunner:
    """Runs ConsistentOnlineDynamicDepth frame by frame on one GPU, eagerly or by graph replay."""

    def __init__(self, estimator, img_metas, use_graph=True, split=None):
        self.est = estimator
        self.metas = img_metas
        self.use_graph = use_graph
        # split = True: the frame is FOUR graphs on three HIP streams (see _capture_split) instead of one graph with
        # parallel branches; default OFF (measured 72 vs 75 frames/s, DESIGN.md finding 14; CODD_SPLIT_GRAPHS=1 or
        # split=True selects it; tests/test_gpu_stereo_net.py::test_split_graphs_equal_single_graph keeps it honest)
        self.split = (os.environ.get("CODD_SPLIT_GRAPHS", "0") == "1") if split is None else split
        self.last = {}  # per-frame side outputs of the last step (e.g. "Ts", the SE3 field: scene-flow evaluation)
        self.state = {}
        self.graph = None
        self._static = None
        self.frames = 0

    def reset(self):
        """New sequence (reference reset_inference_state, model/codd.py:400-433)."""
        self.state = {}
        self.frames = 0
        if self._static is not None:
            self._static["primed"] = False

    # ---- eager ------------------------------------------------------------------------------
    def _eager(self, left, right):
        out = self.est.consistent_online_depth_estimation(left, right, self.metas, self.state)
        self.last = {k: out[k] for k in ("Ts",) if out.get(k) is not None}
        return out["pred_disp"]

    # ---- graph ------------------------------------------------------------------------------
    def _state_tensors(self, state):
        mem = state["memory"]
        return [mem[0], mem[1], mem[2], state["raft_feat"], state["raft_netinp"]]

    def _capture(self, left, right):
        dev = left.device
        st = dict(l=torch.empty_like(left), r=torch.empty_like(right), primed=True)
        st["state"] = [torch.empty_like(t).contiguous() for t in self._state_tensors(self.state)]
        for dst, src in zip(st["state"], self._state_tensors(self.state)):
            dst.copy_(src)
        st["l"].copy_(left)
        st["r"].copy_(right)
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()

        def body():
            s = st["state"]
            state = dict(memory=[s[0], s[1], s[2]], raft_feat=s[3], raft_netinp=s[4])
            out = self.est.consistent_online_depth_estimation(st["l"], st["r"], self.metas, state)
            # state write-back with a KERNEL, not Tensor.copy_: under capture copy_/memset become
            # memcpy/memset graph nodes, and on ROCm 7.2 a captured hipMemsetAsync node was observed
            # to race with the neighbouring kernel nodes (GPU page faults after ~45 replays when
            # eager work ran between replays).  Only kernel nodes are used inside the frame graph.
            ops.copy_many([(dst, src.contiguous()) for dst, src in zip(s, self._state_tensors(state))])
            st["last"] = {k: out[k] for k in ("Ts",) if out.get(k) is not None}  # (lives in the graph's memory pool)
            return out["pred_disp"]

        saved = [t.clone() for t in st["state"]]
        with torch.cuda.stream(stream):
            body()  # warm-up on the side stream (weight packing, allocator)
            for dst, src in zip(st["state"], saved):
                dst.copy_(src)
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, stream=stream):
                st["out"] = body()
        torch.cuda.current_stream(dev).wait_stream(stream)
        # capture does not execute: restore the pre-capture state and replay once for this frame
        for dst, src in zip(st["state"], saved):
            dst.copy_(src)
        self.graph, self._static = g, st

    # ---- split graphs -----------------------------------------------------------------------
    # One captured graph with parallel branches is replayed by the HIP runtime almost in capture order: the image-only
    # branches (RAFT3D feature encoder ~100 launches, context network ~200) and the stereo network share the device
    # only at their seams (rocprofv3: 11.2 of 14.3 busy ms with exactly one kernel resident, the stereo network's first
    # launch 1.0 ms after the frame starts).  Separate graphs launched on separate streams land on separate hardware
    # queues and run side by side:
    #     s1: [fnet]            -> fmap
    #     s2: [cnet]            -> netinp (only the NEXT frame reads it: reference raft3d.py:278)
    #     s0: [stereo] , wait s1, [motion + fusion + state write-back] , wait s2, netinp -> state
    # Same kernels, same arguments, same per-chain order as the single graph: results are identical.
    def _capture_split(self, left, right):
        dev = left.device
        raft = self.est.motion.raft3d
        st = dict(l=torch.empty_like(left), r=torch.empty_like(right), primed=True, split=True)
        st["state"] = [torch.empty_like(t).contiguous() for t in self._state_tensors(self.state)]
        for dst, src in zip(st["state"], self._state_tensors(self.state)):
            dst.copy_(src)
        st["l"].copy_(left)
        st["r"].copy_(right)
        saved = [t.clone() for t in st["state"]]
        s0, s1, s2 = (torch.cuda.Stream(device=dev) for _ in range(3))
        st["streams"] = (s0, s1, s2)
        cur = torch.cuda.current_stream(dev)
        for s in (s0, s1, s2):
            s.wait_stream(cur)

        def stage_b():
            s = st["state"]
            state = dict(memory=[s[0], s[1], s[2]], raft_feat=s[3], raft_netinp=s[4])
            outputs = dict(st["stereo"])
            raft._pending, raft._nowait = dict(fmap=st["fmap"], netinp=st["netinp"]), True
            try:
                self.est.motion(state, outputs, img_metas=self.metas, train_mode=False)
                self.est.fusion.memory_query(outputs, state, img_metas=self.metas)
                self.est.fusion.memory_update(outputs, state, img_metas=self.metas)
            finally:
                raft._pending, raft._nowait = None, False
            # state write-back with kernels (see _capture); netinp is written by step() once s2 has finished
            for dst, src in zip(s[:4], self._state_tensors(state)[:4]):
                ops.add_relu(src.contiguous(), None, relu=False, out=dst)
            st["last"] = {k: outputs[k] for k in ("Ts",) if outputs.get(k) is not None}
            return outputs["pred_disp"]

        def capture(stream, fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(stream):
                out = fn()  # warm-up on the capture stream (weight packing, allocator)
                torch.cuda.synchronize(dev)
                with torch.cuda.graph(g, stream=stream):
                    out = fn()
            return g, out

        with torch.no_grad():
            st["g_f"], st["fmap"] = capture(s1, lambda: raft.fnet(st["l"]))
            st["g_c"], st["netinp"] = capture(s2, lambda: raft.context(st["l"]))
            st["g_s"], st["stereo"] = capture(s0, lambda: self.est.stereo.stereo_matching(st["l"], st["r"], self.metas, {}))
            # (capture does not execute: run the three once so that stage B's warm-up reads real tensors)
            for g in (st["g_f"], st["g_c"], st["g_s"]):
                g.replay()
            torch.cuda.synchronize(dev)
            st["g_m"], st["out"] = capture(s0, stage_b)
        torch.cuda.synchronize(dev)
        for dst, src in zip(st["state"], saved):  # the warm-up advanced the state: restore, step() replays this frame
            dst.copy_(src)
        cur.wait_stream(s0)
        self.graph, self._static = st["g_m"], st

    def _replay_split(self):
        st = self._static
        dev = st["l"].device
        s0, s1, s2 = st["streams"]
        cur = torch.cuda.current_stream(dev)
        for s in (s0, s1, s2):
            s.wait_stream(cur)  # the input copies
        with torch.cuda.stream(s0):
            st["g_s"].replay()
        with torch.cuda.stream(s1):
            st["g_f"].replay()
        with torch.cuda.stream(s2):
            st["g_c"].replay()
        with torch.cuda.stream(s0):
            s0.wait_stream(s1)
            st["g_m"].replay()
            s0.wait_stream(s2)
            ops.add_relu(st["netinp"], None, relu=False, out=st["state"][4])
        cur.wait_stream(s0)

    def eager_frame_on_static_state(self, left, right):
        """One eager (un-captured) steady-state frame on the current recurrent state -- used by
        bench.py to bracket individual launches with events.  Advances the state."""
        if self.est.motion is None and self.est.fusion is None:  # stereo-only estimator: no recurrent state
            return self._eager(left, right)
        if self._static is not None and self._static.get("primed"):
            s = self._static["state"]
            state = dict(memory=[s[0], s[1], s[2]], raft_feat=s[3], raft_netinp=s[4])
        else:
            state = self.state
        out = self.est.consistent_online_depth_estimation(left, right, self.metas, state)
        if state is not self.state:
            for dst, src in zip(self._static["state"], self._state_tensors(state)):
                dst.copy_(src)
        return out["pred_disp"]

    def _capture_stateless(self, left, right):
        """Stereo-only estimator (no motion / fusion, BASELINE.json configs[1]): no recurrent state."""
        dev = left.device
        st = dict(l=left.clone(), r=right.clone(), primed=True, state=[])
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            self._eager(st["l"], st["r"])  # warm-up (weight packing, allocator)
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, stream=stream):
                st["out"] = self._eager(st["l"], st["r"])
        torch.cuda.current_stream(dev).wait_stream(stream)
        self.graph, self._static = g, st

    def step(self, left, right):
        """One frame.  Returns the (fused) disparity [B,1,H,W]; valid until the next call."""
        self.frames += 1
        if self.use_graph and self.est.motion is None and self.est.fusion is None:
            if self.graph is None:
                self._capture_stateless(left, right)
            self._static["l"].copy_(left, non_blocking=True)
            self._static["r"].copy_(right, non_blocking=True)
            self.graph.replay()
            return self._static["out"]
        has_mem = "memory" in self.state or (self._static is not None and self._static.get("primed"))
        if not self.use_graph or not has_mem:
            d = self._eager(left, right)
            return d
        if self.graph is None:
            if self.split and not ops.Fork.serial:
                self._capture_split(left, right)
            else:
                self._capture(left, right)
        elif not self._static["primed"]:
            for dst, src in zip(self._static["state"], self._state_tensors(self.state)):
                dst.copy_(src)
            self._static["primed"] = True
        st = self._static
        st["l"].copy_(left, non_blocking=True)
        st["r"].copy_(right, non_blocking=True)
        if st.get("split"):
            self._replay_split()
        else:
            self.graph.replay()
        self.state = {"memory": True}  # state now lives in the static buffers
        self.last = st.get("last", {})
        return st["out"]


class PipelinedRunner:
    """Frame pipeline of depth two: while the motion + fusion stages of frame t run, the image-only work
    of frame t+1 -- the whole stereo network, RAFT3D's feature encoder and the context network (reference
    hitnet.py:75-100, raft3d.py:151-160: none of them reads the recurrent state) -- runs beside them on
    side streams of the SAME graph.  The GRU loop's 72x120 launches leave CUs idle (576 workgroups on 256
    CUs); the next frame's 16-channel full-resolution layers fill them.  Results are identical to
    FrameRunner's (same kernels, same per-stream order); the price is one frame of latency:

        push(l0, r0) -> None;  push(l1, r1) -> disparity of frame 0;  ...;  flush() -> last frame.

    Stage A (frame t+1) writes fresh tensors; stage B (frame t) reads the persistent copies ``a_cur`` of
    stage A's previous results; the tail of the graph writes the recurrent state back and rotates A."""

    A_KEYS = ("pred_disp", "left_feat", "right_feat", "left_img", "fmap", "netinp")

    def __init__(self, estimator, img_metas, use_graph=True):
        if estimator.motion is None or estimator.fusion is None:
            raise ValueError("the frame pipeline needs the motion and fusion stages (use FrameRunner)")
        self.est, self.metas, self.use_graph = estimator, img_metas, use_graph
        self.reset()
        self.graph = None
        self._static = None

    def reset(self):
        self.state = {}
        self.a_cur = None  # stage-A results of the frame waiting for its stage B
        self.pushed = 0
        if getattr(self, "_static", None) is not None:
            self._static["primed"] = False

    # ---- stages -------------------------------------------------------------------------------
    def _stage_a(self, left, right, side):
        """image-only work on ``side`` (+ RAFT3D's own two side streams); returns the new A dict."""
        dev = left.device
        raft = self.est.motion.raft3d
        side.wait_stream(torch.cuda.current_stream(dev))
        # stage A runs on a side stream of the captured graph: HRNet's branch streams and the stereo network's second
        # stream could only be forked from that (already forked) stream, which hipGraphInstantiate of ROCm 7.2 does not
        # survive -> no inner forks, for THIS call only (the estimator is shared with FrameRunner / eager callers,
        # whose schedule and runner-cache key must not change behind their back)
        cn = getattr(raft, "cnet", None)
        scoped = [(m, a, getattr(m, a, True)) for m, a in
                  (((cn[0], "fork_branches"),) if cn is not None and hasattr(cn[0], "fork") else ()) +
                  (((self.est.stereo, "fork_streams"),) if self.est.stereo is not None else ())]
        for m, a, _ in scoped:
            setattr(m, a, False)
        try:
            with torch.cuda.stream(side):
                self.est.motion.prefetch(left)
                pend = raft._pending
                raft._pending = None
                if pend is None:
                    raise RuntimeError("PipelinedRunner needs the side streams (ops.Fork.serial must be False)")
                out = self.est.stereo.stereo_matching(left, right, self.metas, {})
        finally:
            for m, a, v in scoped:
                setattr(m, a, v)
        return dict(pred_disp=out["pred_disp"], left_feat=out["left_feat"], right_feat=out["right_feat"],
                    left_img=left, fmap=pend["fmap"], netinp=pend["netinp"])

    def _join_a(self, side, dev):
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(side)
        for s in self.est.motion.raft3d._side:
            cur.wait_stream(s)

    def _stage_b(self, a, state):
        """motion + fusion of the frame whose stage-A results are ``a`` (reference model/codd.py:103-121)."""
        raft = self.est.motion.raft3d
        outputs = dict(pred_disp=a["pred_disp"], left_feat=a["left_feat"], right_feat=a["right_feat"],
                       left_img=a["left_img"])
        raft._pending, raft._nowait = dict(fmap=a["fmap"], netinp=a["netinp"]), True
        try:
            self.est.motion(state, outputs, img_metas=self.metas, train_mode=False)
            self.est.fusion.memory_query(outputs, state, img_metas=self.metas)
            self.est.fusion.memory_update(outputs, state, img_metas=self.metas)
        finally:
            raft._pending, raft._nowait = None, False
        return outputs["pred_disp"]

    @staticmethod
    def _state_tensors(state):
        mem = state["memory"]
        return [mem[0], mem[1], mem[2], state["raft_feat"], state["raft_netinp"]]

    def _side_stream(self, dev):
        if getattr(self, "_side", None) is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    # ---- eager frames (the first two pushes, or use_graph = False) -------------------------------
    def _eager_step(self, left, right):
        dev = left.device
        side = self._side_stream(dev)
        new = self._stage_a(left, right, side)
        pred = self._stage_b(self.a_cur, self.state) if self.a_cur is not None else None
        self._join_a(side, dev)
        new["left_img"] = left.clone()  # the caller may reuse its image buffer
        self.a_cur = new
        return pred

    # ---- graph ------------------------------------------------------------------------------------
    def _capture(self, left, right):
        dev = left.device
        st = dict(l=torch.empty_like(left), r=torch.empty_like(right), primed=True)
        st["state"] = [t.clone().contiguous() for t in self._state_tensors(self.state)]
        st["a"] = {k: self.a_cur[k].clone().contiguous() for k in self.A_KEYS}
        st["l"].copy_(left)
        st["r"].copy_(right)
        stream, side = torch.cuda.Stream(device=dev), self._side_stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()

        def body():
            s, a = st["state"], st["a"]
            state = dict(memory=[s[0], s[1], s[2]], raft_feat=s[3], raft_netinp=s[4])
            new = self._stage_a(st["l"], st["r"], side)
            pred = self._stage_b(a, state)
            self._join_a(side, dev)
            # tail (kernel nodes only, see FrameRunner._capture): recurrent state, then rotate stage A
            for dst, src in zip(s, self._state_tensors(state)):
                ops.add_relu(src.contiguous(), None, relu=False, out=dst)
            for k in self.A_KEYS:
                ops.add_relu(new[k].contiguous(), None, relu=False, out=a[k])
            return pred

        saved = [t.clone() for t in st["state"]] + [st["a"][k].clone() for k in self.A_KEYS]

        def restore():
            for dst, src in zip(st["state"] + [st["a"][k] for k in self.A_KEYS], saved):
                dst.copy_(src)

        with torch.cuda.stream(stream):
            body()  # warm-up (weight packing, allocator)
            restore()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, stream=stream):
                st["out"] = body()
        torch.cuda.current_stream(dev).wait_stream(stream)
        restore()
        self.graph, self._static = g, st

    def push(self, left, right):
        """Feed frame k; returns the fused disparity [B,1,H,W] of frame k-1 (None for k = 0); the tensor is
        valid until the next call."""
        self.pushed += 1
        if self.pushed <= 2 or not self.use_graph:
            return self._eager_step(left, right)
        if self.graph is None:
            self._capture(left, right)
        elif not self._static["primed"]:
            for dst, src in zip(self._static["state"], self._state_tensors(self.state)):
                dst.copy_(src)
            for k in self.A_KEYS:
                self._static["a"][k].copy_(self.a_cur[k])
            self._static["primed"] = True
        st = self._static
        st["l"].copy_(left, non_blocking=True)
        st["r"].copy_(right, non_blocking=True)
        self.graph.replay()
        return st["out"]

    def flush(self):
        """Disparity of the last pushed frame (runs its motion + fusion stages; ends the sequence)."""
        if self.pushed == 0:
            return None
        if self.pushed <= 2 or not self.use_graph:
            a, state = self.a_cur, self.state
        else:
            s, a = self._static["state"], self._static["a"]
            state = dict(memory=[s[0], s[1], s[2]], raft_feat=s[3], raft_netinp=s[4])
        pred = self._stage_b(a, state)
        self.reset()
        return pred
