"""Several controller batches of different Models stepped in the same loop on one GPU — the batched form of the
reference's multiple_controller example (multiple_controller/main.cpp:89-118: two Cgmres objects of different Model
types, `controller1.control(u1,x1); controller2.control(u2,x2);` back to back, one timer around both).

Each member batch has its own handle and HIP stream, so the tick kernels of the different models overlap on the
device; `control_device` enqueues all of them and `synchronize` joins them.  Members are created on the "wg-lean"
mapping (variant 3: half a CU's LDS per workgroup) where their sizes allow it, so workgroups of DIFFERENT members
share a CU and the models really run side by side instead of taking turns on the CUs."""
from . import CgmresBatch, CgmresHipError


def _cu_count(device):
    try:
        import torch
        return int(torch.cuda.get_device_properties(device).multi_processor_count)
    except Exception:
        return 256  # MI355X


class MultipleController:
    def __init__(self, specs, device=0, streams=None):
        """specs: list of dicts of CgmresBatch keyword arguments (model, batch, dv, k_max, ...).
        streams: optional list of hipStream_t values (e.g. torch.cuda.Stream().cuda_stream), one per member."""
        self.members = []
        self._streams = None
        if streams is None and len(specs) > 1:
            # One HIP stream per member, with ALTERNATING priorities: streams of one priority can be mapped onto the
            # same hardware queue, and the members' kernels then run one after the other instead of side by side
            # (measured: the joint tick of config 4 is bimodal, 245 us or 411 us = the sum of the two).  Streams of
            # different priority never share a queue.  Needs torch for the stream objects; without it the library
            # creates the streams itself (same priority).
            try:
                import torch
                lo, hi = torch.cuda.Stream.priority_range()
                self._streams = [torch.cuda.Stream(device=device, priority=(hi if i % 2 else lo))
                                 for i in range(len(specs))]
                streams = [st.cuda_stream for st in self._streams]
            except Exception:
                streams = None
        # Sharing CUs pays when the members together need more 16-instance workgroups than the GPU has CUs; below that
        # every workgroup gets a CU of its own and the faster one-workgroup-per-CU mapping is kept (e.g. the 8-GPU
        # shards of multiple_controller: 2 x 32 workgroups per GPU).
        share_cus = len(specs) > 1 and sum(-(-int(kw.get("batch", 1)) // 16) for kw in specs) > _cu_count(device)
        for i, kw in enumerate(specs):
            kw = dict(kw)
            kw.setdefault("device", device)
            if streams is not None:
                kw["stream"] = streams[i]
            if share_cus and "variant" not in kw:
                try:
                    self.members.append(CgmresBatch(variant=3, **kw))
                    continue
                except CgmresHipError:
                    pass  # sizes outside the lean plan: the library's default mapping
            self.members.append(CgmresBatch(**kw))

    def __len__(self):
        return len(self.members)

    def __getitem__(self, i):
        return self.members[i]

    def control(self, xs):
        """Host arrays in, host arrays out: [u_1, u_2, ...] for [x_1, x_2, ...]."""
        return [m.control(x) for m, x in zip(self.members, xs)]

    def control_device(self, us, xs):
        for m, u, x in zip(self.members, us, xs):
            m.control_device(u, x)

    def closed_loop_device(self, xs, us, n_ticks):
        """n_ticks of every member's closed loop; ticks of different members interleave on their streams."""
        from . import TICKS_PER_LAUNCH  # one launch advances this many ticks of a member: interleave launch-wise
        done = 0
        while done < n_ticks:
            n = min(TICKS_PER_LAUNCH, n_ticks - done)
            for m, x, u in zip(self.members, xs, us):
                m.closed_loop_device(x, u, n)
            done += n

    def synchronize(self):
        for m in self.members:
            m.synchronize()

    def close(self):
        for m in self.members:
            m.close()
