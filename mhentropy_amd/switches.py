"""The ONE table of run-time switches (environment variables read once per process by the host code or by libmhe_hip.so).

The defaults ARE the product path: what bench.py times and what the parity tests run.  Every alternative is a measurement switch - it
selects another HIP kernel or another ordering of the same launches so that an A/B runs inside one `gpurun` call (box-to-box spread
exceeds most single changes); none selects a CPU path.  tests/test_abi_and_tables.py checks that this table and the source agree (every
variable the source reads is listed here with the default the source uses, and nothing else); bench.py prints `config.switches_set` =
non_default(), which is empty on the driver's run.

Compile-time knobs (not environment variables: tuning builds only) are MHE_HALO_ABL, MHE_P8_ABLATIONS, MHE_CONV_XCD_ORDER,
MHE_CONV_BN_EPILOGUE (csrc/*.hip, *.h) and MHE_EXTRA_FLAGS (build.py: extra hipcc flags)."""
import os

# name: (default, where it is read, what the non-default value selects)
SWITCHES = {
    # ---- forward trunk (resnet.py / csrc/conv*.hip)
    "MHE_CONV_TILE": ("-1", "csrc/conv.hip", "force convolution variant k where the geometry admits it (tuning runs)"),
    "MHE_CONV_STREAM": ("1", "csrc/conv.hip", "0: no streaming 1x1 / row-streaming 3x3 kernels (variants 8 / 9)"),
    "MHE_CONV_WIDE": ("1", "csrc/conv.hip", "0: no resident-slab kernel (variant 11)"),
    "MHE_CONV_TAIL": ("1", "csrc/conv.hip", "0: no residual-tail kernel with transfer waves (variant 10)"),
    "MHE_CONV_P8H": ("1", "csrc/conv.hip", "0: no 256x128 phase-pipelined tile (variant 13)"),
    "MHE_CONV_HALO": ("1", "resnet.py, train.py", "0: 3x3 stride-1 units of layer2 / layer3 on the im2col kernels instead of the resident-tile kernel"),
    "MHE_BN_APPLY": ("pass", "resnet.py", "pass | load: where a producer's BatchNorm + ReLU is applied by default"),
    "MHE_BN_APPLY_1X1": ("auto", "resnet.py", "pass | load | auto for the 1x1 consumers"),
    "MHE_BN_APPLY_3X3": ("auto", "resnet.py", "pass | load | auto for the 3x3 consumers"),
    "MHE_BN_LOAD_S2": ("256", "resnet.py", "BatchNorm on the load of the stride-2 3x3 launches up to this many channels"),
    "MHE_FUSE_TAIL": ("1", "resnet.py", "0: block tails as separate bn_act passes"),
    "MHE_FUSE_RECOMPUTE": ("1", "resnet.py", "0: conv3 of layer1 / layer2 written and read back (forward-only path)"),
    "MHE_RECOMPUTE_STATS": ("gram", "resnet.py", "stream: bn3 statistics from the statistics-only launch instead of the Gram matrix"),
    "MHE_STEM_POOL": ("1", "resnet.py", "0: stem convolution and max pool as two kernels"),
    "MHE_FUSE_POOL": ("1", "resnet.py", "0: last block tail and average pool as two launches"),
    "MHE_GRAM_WGS_64": ("512", "csrc/conv_gram.hip", "workgroups of the 64-channel Gram launch"),
    "MHE_GRAM_WGS_128": ("256", "csrc/conv_gram.hip", "workgroups of the 128-channel Gram launch"),
    # ---- flow / MANO
    "MHE_FLOW_FRAG": ("1", "flows.py, train.py", "0: second-generation coupling-stack kernel (flow_ns.hip) instead of the fragment-streaming one"),
    "MHE_FLOW_W1_SETS": ("2", "csrc/flow_fwd.hip", "3: a third register set of layer-1 weight fragments (measured equal)"),
    "MHE_GLOW_FUSED": ("1", "glow.py, train_glow.py", "0: the Glow branch's sampling pass layer by layer (~60 launches) instead of the one-launch kernel"),
    "MHE_GLOW_REV_FUSED": ("1", "train_glow.py", "0: the Glow branch's reverse pass over the tape stage by stage instead of the one-launch chain (csrc/glow_rev.hip)"),
    "MHE_MANO_FOUR": ("1", "csrc/mano.hip", "0: one hypothesis per wavefront"),
    "MHE_LBS_MFMA": ("1", "body.py", "0: the body model's skinning with one thread per vertex (csrc/body.hip) instead of the matrix-core kernel (csrc/lbs_skin.hip)"),
    "MHE_MANO_SKIN_MFMA": ("1", "csrc/mano.hip", "0: full-mesh skinning with one thread per vertex (round 1) instead of the matrix-core kernel on bf16 pieces (csrc/mano_skin.hip)"),
    # ---- train step (train.py / csrc/wgrad.hip, trunk_bwd.hip)
    "MHE_TRAIN_RECOMPUTE": ("1", "train.py", "0: conv3 of layer1 / layer2 written by the train step's forward pass"),
    "MHE_CONV3_FOLD": ("1", "train.py", "0: conv3 + bn3 reversed by reading y3 (no Gram fold)"),
    "MHE_CONV3_FOLD_CAT": ("1", "train.py", "0: the fold's second product as a launch of its own"),
    "MHE_SHORTCUT_FOLD": ("1", "train.py", "0: layer1's shortcut reversed by the apply pass"),
    "MHE_STEM_BWD_TWO_PASS": ("1", "train.py", "0: the stem's scattered pool gradient written and read back"),
    "MHE_STEM_POOL_FUSED": ("1", "train.py", "0: stem BatchNorm + ReLU not folded into its max pool"),
    "MHE_STEM_POOLED_SUMS": ("1", "train.py", "0: the stem's BatchNorm-reverse sums from a walk over its full-resolution output"),
    "MHE_STEM_WGRAD_PAIRS": ("1", "train.py", "0: the stem's weight gradient over single pixels (3 channels padded to 8)"),
    "MHE_GATE_BITS": ("1", "train.py", "0: ReLU gates read from the block-wide tensors instead of their bits"),
    "MHE_CONV_HALO_DG": ("1", "train.py", "0: 3x3 data gradients on the im2col kernels"),
    "MHE_HALO_BN_ON_LOAD": ("0", "train.py", "1: the unit's BatchNorm reverse on the resident-tile data-gradient launch's load (measured equal)"),
    "MHE_BN_REDUCE_FUSED": ("1", "train.py", "0: BatchNorm-reverse sums in a pass of their own"),
    "MHE_BN_BWD_ON_LOAD": ("1", "train.py", "0: BatchNorm-reverse apply always as a pass"),
    "MHE_BN_BWD_ON_LOAD_MAXC": ("128", "train.py", "widest bottleneck whose conv3 reverse takes the apply on its operand load"),
    "MHE_BN_BWD_ON_LOAD_WIDE": ("1", "train.py", "0: layer3 / layer4's conv3 reverse without the apply on its load"),
    "MHE_BN_BWD_APPLY_WIDE": ("1", "csrc/trunk_bwd.hip", "0: 8-byte lanes in the apply pass"),
    "MHE_COND_BWD_BF16": ("1", "train.py", "0: the conditioning projections' reverse products on f32 operands"),
    "MHE_FLOW_WGRAD_GROUPED": ("1", "train.py", "0: one weight-gradient launch per coupling net"),
    "MHE_FLOW_REV_FUSED": ("1", "train.py", "0: the flow's reverse chain coupling by coupling"),
    "MHE_FLOW_RECOMPUTE": ("0", "train.py", "1: the reverse pass re-evaluates the flow nets instead of reading emitted activations"),
    "MHE_LAZY_FALLBACK_TABLES": ("1", "train.py", "0: the fallback operand layouts refreshed every step"),
    "MHE_POISON_STALE_TABLES": ("0", "train.py", "1 (debug): fallback operand layouts a repack leaves behind are filled with NaN"),
    "MHE_GATHER_AFFINE": ("1", "train.py", "0: the bf16 operand re-pack from one index per element instead of (base, stride, validity) per eight"),
    "MHE_WGRAD_MULTI": ("1", "train.py", "0: one weight-gradient launch per trunk layer instead of one multi-problem launch per gradient bucket and tile shape"),
    "MHE_WGRAD_MULTI_WGS": ("2048", "csrc/wgrad.hip", "workgroups a multi-problem launch of the 4-wave tiles aims at"),
    "MHE_WGRAD_MULTI_WGS_BIG": ("768", "csrc/wgrad.hip", "workgroups a multi-problem launch of the 256 x 256 tile aims at"),
    "MHE_WGRAD_W16": ("1", "csrc/wgrad.hip", "0: the 256 x 256 weight-gradient tile on eight waves of 64 x 128 instead of sixteen of 64 x 64"),
    "MHE_WGRAD_DMA": ("1", "csrc/wgrad.hip", "0: register-staged bf16 weight-gradient kernel"),
    "MHE_WGRAD_BIG": ("1", "csrc/wgrad.hip", "0: no 256x256 weight-gradient tile"),
    "MHE_WGRAD_XCD": ("1", "csrc/wgrad.hip", "0: weight-gradient tiles in grid order"),
    "MHE_WGRAD_WGS": ("512", "csrc/wgrad.hip", "target workgroup count of a split launch"),
    "MHE_WGRAD_MINCHUNK": ("512", "csrc/wgrad.hip", "shortest pixel slice of the bf16 kernels"),
    "MHE_WGRAD_SLABS": ("all", "csrc/wgrad.hip", "big: small layers add their tiles with f32 atomics (order-dependent sums)"),
    "MHE_WGRAD_F32MFMA": ("", "csrc/wgrad.hip", "set: bf16 operands through the f32 MFMA kernel"),
    # ---- processes
    "MHE_GRAD_EXCHANGE": ("f32", "dist.py", "bf16: gradient buckets exchanged as bf16 (all-to-all + all-gather, f32 accumulation) instead of an f32 all-reduce"),
    "MHE_DIST_FORCE": ("0", "dist.py", "1: issue every collective in a group of ONE rank (one-GPU rehearsal of the RCCL path)"),
    "MHE_BENCH_REHEARSE": ("0", "bench.py", "1: all ranks share cuda:0 and talk over gloo (development only)"),
}


def non_default(environ=None):
    """{name: value} of the switches set in the environment to something other than their default"""
    env = os.environ if environ is None else environ
    return {k: env[k] for k, (default, _, _) in SWITCHES.items() if k in env and env[k] != default}


def markdown_table():
    rows = ["| variable | default | read by | non-default value selects |", "|---|---|---|---|"]
    rows += [f"| `{k}` | `{d or '(unset)'}` | `{w}` | {m} |" for k, (d, w, m) in SWITCHES.items()]
    return "\n".join(rows)


if __name__ == "__main__":
    print(markdown_table())
