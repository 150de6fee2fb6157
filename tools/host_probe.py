#!/usr/bin/env python
"""How far ahead of the GPU does the host run?  C2 training steps through bench.py's own model / batches: per step the
host time spent INSIDE optimize_parameters (enqueue only, nothing waits for the device) next to the device time per step.
host << device: the launches of a step are queued long before the GPU reaches them (no launch-bound stretch in the step);
host ~ device: the host is the limiter somewhere.     python tools/host_probe.py [steps] [hold_ms]

hold_ms > 0 (for runs UNDER rocprofv3, whose per-dispatch interception makes the host 2.8x slower -- 56 instead of 20 ms per
step, i.e. slower than the device: a traced step is launch-starved and its gaps are the profiler's): every stream of the
schedule first waits behind a spin kernel of hold_ms on the main stream, the host enqueues ALL the steps meanwhile, and
the GPU then runs them with every launch already queued -- the trace shows the schedule of the untraced run."""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from neurips18_hierchical_image_manipulation_amd import synth
    from neurips18_hierchical_image_manipulation_amd.models import create_model
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    wl = bench.WORKLOADS['c2']
    dev = torch.device('cuda', 0)
    model = create_model(dict(wl['flags'], gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/him_bench', name='probe',
                              batchSize=wl['bs']))
    model.netG.load_state_dict(synth.init_state_dict(model.netG.state_dict(), 1))
    model.netD.load_state_dict(synth.init_state_dict(model.netD.state_dict(), 2))
    batches = [{k: v.to(dev) for k, v in synth.make_batch(s, 0, wl['bs'], wl['H'], wl['W'], wl['label_nc'], wl['color']).items()}
               for s in range(4)]
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))
    for b in batches:
        b['ready_event'] = ready
    for i in range(4):
        model.optimize_parameters(batches[i % 4])
    torch.cuda.synchronize()
    hold_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    held = None
    if hold_ms > 0:
        from neurips18_hierchical_image_manipulation_amd import ops
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.cuda._sleep(20_000_000)
        b.record()
        b.synchronize()
        per_ms = 20_000_000 / a.elapsed_time(b)
        main = torch.cuda.current_stream(dev)
        torch.cuda.synchronize()
        ref0 = torch.cuda.Event(enable_timing=True)
        ref0.record()                                    # executes at once: device time 0 == host time t_ref
        t_ref = time.perf_counter()
        torch.cuda._sleep(int(hold_ms * per_ms))
        gate = torch.cuda.Event()
        gate.record(main)
        for st in (ops._side_stream(dev), ops._opt_stream(dev), ops._d_opt_stream(dev), ops._vgg_stream(dev),
                   ops._real_stream(dev)):
            st.wait_event(gate)
        ready2 = torch.cuda.Event()
        ready2.record(main)
        for bt in batches:
            bt['ready_event'] = ready2          # the input encoding waits for the gate too
        held = dict(hold_ms=hold_ms, spin_cycles_per_ms=round(per_ms), ref=ref0, t_ref=t_ref)
    host, lag, marks = [], [], []
    ref = torch.cuda.Event(enable_timing=True)
    if not held:
        ref.record()
    else:
        ref = held.pop('ref')
    t0 = held.pop('t_ref') if held else time.perf_counter()
    for i in range(steps):
        a = time.perf_counter()
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[-1].record()                               # reached by the main stream when the device starts step i
        model.optimize_parameters(batches[i % 4])
        host.append((time.perf_counter() - a) * 1e3)
        lag.append((time.perf_counter() - t0) * 1e3)      # wall clock at which step i was fully enqueued
    t_enq = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3
    dev_start = [round(ref.elapsed_time(m), 1) for m in marks]
    # a step is launch-starved when the device reached it before the host had finished enqueuing it
    starved = [i for i in range(steps) if lag[i] > dev_start[i]]
    print(json.dumps(dict(steps=steps, device_reached_step_at_ms=dev_start, launch_starved_steps=starved, host_ms_per_step=[round(h, 2) for h in host], enqueue_done_at_ms=[round(l, 1) for l in lag],
                          all_enqueued_ms=round(t_enq, 1), device_done_ms=round(t_all, 1),
                          device_ms_per_step=round((t_all - dev_start[0]) / steps, 2),
                          host_ms_per_step_mean=round(sum(host) / steps, 2), held=held)))


if __name__ == '__main__':
    main()
