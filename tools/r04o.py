import sys, os, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import test_model_gpu as T
from neurips18_hierchical_image_manipulation_amd import config, ops, _cabi as cb
def run(name, sched, algo):
    try:
        with config.schedule(**sched), ops.algo_scope(**algo):
            T.PARITY_K_TYPICAL_WINOGRAD = 1e9
            try:
                T._teacher_forced('tiny_twostream', 6, k_typical=1e9, out_tag='probe_tw')
            except AssertionError as e:
                print(name, 'assert', str(e)[:100])
        d = json.load(open(os.path.join(ROOT, 'gpurun_out', 'teacher_forced_probe_tw.json')))
        g = d['grad_distance_from_fp64']; i = g['tensors'].index('G/decoder.0.weight')
        print('%-22s hip %s | orc %s' % (name, ' '.join('%.0e' % r[i] for r in g['hip']), ' '.join('%.0e' % r[i] for r in g['oracle_live'])), flush=True)
    except Exception as e:
        print(name, 'ERR', repr(e)[:200])
run('default', {}, {})
run('default again', {}, {})
run('no lincomb', dict(lincomb=False), {})
run('serial', dict(config.SERIAL), {})
run('no panel cache', dict(panel_cache=False), {})
run('no dead bias skip', dict(dead_bias_skip=False), {})
run('direct form', {}, dict(wino_min_c=-1))
run('no split-K', {}, dict(disable=cb.ALGO_NO_SPLITK))
run('generic', {}, dict(disable=cb.ALGO_GENERIC_CONV))
