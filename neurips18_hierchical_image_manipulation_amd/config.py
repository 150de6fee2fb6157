"""Host-side switches of the training step's SCHEDULE and of optional fusions -- one runtime object instead of module
constants latched from the environment at import.

``SCHED`` holds the shipped defaults; the ``HIM_*`` environment variables below are read ONCE, here, when the package is
imported (tools/ A/B runs), and everything can be changed at run time -- ``with config.schedule(wgrad_stream=False): ...``
-- which is how the parity suite compares the multi-stream schedule bit for bit with the serial one
(tests/test_model_gpu.py::test_multi_stream_schedule_is_bit_identical_to_the_serial_one).  None of these switches changes
the arithmetic of a step; the kernel-selection switches, which do, travel in ``HimAlgo`` (ops.current_algo)."""
import contextlib
import os


def _env(name, default):
    v = os.environ.get(name)
    return default if v is None else v != '0'


class Schedule(object):
    # name -> (environment variable, default, what it does)
    FIELDS = {
        'wgrad_stream': ('HIM_WGRAD_STREAM', True, 'weight gradients on a side stream next to the data-gradient chain'),
        'd_wgrad_routes': ('HIM_D_WGRAD_ROUTES', True, "loss_D.backward(): D's weight gradients on the VGG stream"),
        'real_first': ('HIM_REAL_FIRST', False, "round 2's issue order: real-image branch enqueued before the generator"),
        'real_ahead': ('HIM_REAL_AHEAD', True, 'D(real) + VGG(real) on their own stream next to the generator forward'),
        'd_backward_first': ('HIM_D_BACKWARD_FIRST', True, 'loss_D.backward() before loss_G.backward()'),
        'vgg_stream': ('HIM_VGG_STREAM', True, 'VGG(fake) forward / backward on its own stream'),
        'vgg_backward_early': ('HIM_VGG_BACKWARD_EARLY', False, "VGG's backward started before loss_D.backward()"),
        'vgg_batched': ('HIM_VGG_BATCHED', False, 'VGG(real) and VGG(fake) as ONE batch-2B forward chain'),
        'd_split_input': ('HIM_D_SPLIT_INPUT', True, 'discriminators take (condition, image) pairs'),
        'd_scale_streams': ('HIM_D_SCALE_STREAMS', False, 'one stream per PatchGAN scale'),
        'share_fake_pass': ('HIM_SHARE_FAKE_PASS', True, 'the fake-image discriminator pass computed once per step'),
        'vgg_gated': ('HIM_VGG_GATED', True, "VGG's ReLU backward folded into the gradient producers"),
        'onehot_stem': ('HIM_ONEHOT_STEM', True, 'generator stem evaluated from the label ids'),
        'label_ids': ('HIM_LABEL_IDS', True, 'encode_input keeps [one-hot | dense] as (id map, dense channels) (ops.LabelCond): the one-hot block is materialised only for consumers that cannot read ids'),
        'd_from_ids': ('HIM_D_FROM_IDS', True, 'first PatchGAN convolution of scale 0 evaluated from the label ids (table lookups + run-length weight gradient); the pooled scales from 3x3 class counts'),
        'panel_cache': ('HIM_PANEL_CACHE', True, 'weight panels cached on the parameter, rebuilt after Adam'),
        'resblock_fused': ('HIM_RESBLOCK_FUSED', False, 'ResnetBlock with the norms inside the Winograd transforms'),
        'dead_bias_skip': ('HIM_DEAD_BIAS_SKIP', True, 'no bias gradient in front of a mean-subtracting norm'),
        'panel_pipeline': ('HIM_PANEL_PIPELINE', False, "the next generator forward waits for G's Adam kernel and, per layer, for that layer's rebuilt weight panel -- not for the whole panel rebuild pass"),
        'lincomb': ('HIM_LINCOMB', True, 'scalar loss arithmetic as one launch (him_lincomb) instead of one-element ATen ops'),
        'd_update_early': ('HIM_D_UPDATE_EARLY', True, "one rank: D's Adam step starts inside loss_G.backward(), as soon as the gradient has passed back through the discriminator (its last reader of the step), instead of after the generator's whole backward -- the next step's real-image branch then starts under the generator's Adam step; with a gradient exchange attached the update stays behind the backward pass"),
        'inputs_on_real_stream': ('HIM_INPUTS_ON_REAL_STREAM', False, "input encoding on the real-image stream: with D updated early, the NEXT step's encoding + D(real) + VGG(real) run under this step's generator backward / Adam instead of behind them"),
        'real_vgg_first': ('HIM_REAL_VGG_FIRST', False, "real-image stream: VGG(real) in front of the wait for D's update and D(real)"),
        'zero_grad_side': ('HIM_ZERO_GRAD_SIDE', True, 'one rank: optimize_parameters() zeroes the gradient arenas on the weight-gradient stream before the forward pass (under it) instead of on the main stream in front of the backward pass; with a gradient exchange attached the fill stays on the main stream'),
        'conv_in_fused': ('HIM_CONV_IN_FUSED', True, 'Conv2d -> InstanceNorm [-> act] blocks through him_conv2d_in_act_fwd: split-K layers hand their slabs to the InstanceNorm kernel (no finish pass)'),
        'adam_chunked': ('HIM_ADAM_CHUNKED', False, "the generator's Adam step + panel rebuild bucket by bucket DURING its backward pass, as each 64 MB gradient bucket becomes final (and, data parallel, has been exchanged), instead of one 5 GB pass behind the last weight gradient"),
        'g_tail_wgrad_alt': ('HIM_G_TAIL_WGRAD_ALT', True, "GlobalGenerator: the weight gradients of the down-convolutions (the last of the backward pass) on the VGG stream -- idle since VGG's backward -- next to the ResnetBlock stack's weight-gradient GEMMs still queued on the weight-gradient stream, instead of behind them (r06g trace: 2.5 ms at the end of the step with one stream working)"),
        'g_head_wgrad_alt': ('HIM_G_HEAD_WGRAD_ALT', True, "GlobalGenerator: the weight gradients of the up-convolutions (the first MFMA weight gradients of the backward pass) on the VGG stream too, so that the weight-gradient stream reaches the ResnetBlock stack together with the data-gradient chain"),
        'adam_chunked_dp': ('HIM_ADAM_CHUNKED_DP', True, "with a gradient EXCHANGE attached (data parallel, or bench.py --fake-comm's stand-in): the generator's Adam step + panel rebuild bucket by bucket, each right behind its bucket's all-reduce on the optimizer stream (round 6 default: measured +0.06 ms per step next to the exchange stand-in against +0.51 ms for one pass behind the last bucket, profiles/r05_ab_log.txt); one rank without an exchange: see adam_chunked"),
        'adam_split_stem': ('HIM_ADAM_SPLIT_STEM', True, "GlobalGenerator: the generator's Adam step + panel rebuild for everything but the stem starts behind the LAST data gradient, next to the stem's run-length weight gradient (0.5 ms, LDS-bound, the last kernel of the backward pass) instead of behind it; the stem's slice follows (FusedAdam.begin_step / step_range / step: bit-identical)"),
        'stem_wgrad_fork': ('HIM_STEM_WGRAD_FORK', True, "one-hot stem convolutions with dense channels: the dense channels' slice of the weight gradient (MFMA) + the bias gradient on the data-gradient stream, next to the label-id slice (run-length kernel, LDS-bound) on the weight-gradient stream (him_conv2d_onehot_bwd_weight_part) instead of behind it: the generator stem's weight gradient is the last kernel chain of the step and what the next generator forward waits for"),
        'd_prefill_cond': ('HIM_D_PREFILL_COND', True, "the condition channels of the first PatchGAN conv's input buffers copied at the start of the step (ops.cond_pyramid), only the image channels per pass"),
        'keep_wino_input': ('HIM_KEEP_WINO_INPUT', True, "forward keeps the Winograd-transformed input for the layer's weight gradient"),
    }
    # negative spellings kept for the recorded A/B command lines of rounds 2-3
    LEGACY_OFF = {'HIM_NO_ONEHOT_STEM': 'onehot_stem', 'HIM_NO_PANEL_CACHE': 'panel_cache',
                  'HIM_DEAD_BIAS_GRAD': 'dead_bias_skip'}

    def __init__(self):
        for k, (env, default, _) in self.FIELDS.items():
            setattr(self, k, _env(env, default))
        for env, k in self.LEGACY_OFF.items():
            if os.environ.get(env) is not None:
                setattr(self, k, False)

    def as_dict(self):
        return {k: bool(getattr(self, k)) for k in self.FIELDS}


SCHED = Schedule()

# every stream of the step switched off: the reference's own order on ONE stream
SERIAL = dict(wgrad_stream=False, d_wgrad_routes=False, real_ahead=False, d_backward_first=False, vgg_stream=False,
              vgg_backward_early=False, d_scale_streams=False, d_update_early=False, inputs_on_real_stream=False,
              zero_grad_side=False, real_vgg_first=False, adam_chunked=False, adam_chunked_dp=False, adam_split_stem=False, g_tail_wgrad_alt=False, g_head_wgrad_alt=False,
              stem_wgrad_fork=False)


@contextlib.contextmanager
def schedule(**fields):
    saved = {k: getattr(SCHED, k) for k in fields}
    for k in fields:
        if k not in Schedule.FIELDS:
            raise KeyError(k)
    try:
        for k, v in fields.items():
            setattr(SCHED, k, bool(v))
        yield SCHED
    finally:
        for k, v in saved.items():
            setattr(SCHED, k, v)
