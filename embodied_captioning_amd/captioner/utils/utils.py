"""Plugin configuration objects - same fields as the reference's ``experimenting_env/captioner/utils/utils.py:2-12``
plus optional keys (defaults keep existing yamls working): num_beams, max_length, dtype, batch_size, device,
image_size (CoCa: open_clip's force_image_size), streams (BLIP: engines / HIP streams the micro-batches of one call rotate
over, engine.EnginePool; 1 = one engine; the BLIP-2 / CoCa wrappers run one engine and warn when asked for more),
coalesce_rows (BLIP with streams > 1: the pool merges consecutive micro-batches into passes of at most that many rows - same
captions, fewer decode chains; None = 4 x batch_size up to 1024, 0 = off),
early_exit_poll (look for "every caption finished" every n decode steps; None = 4), max_new_tokens (BLIP-2: tokens to
generate, HF's name; None = 20 as HF's generate default - `max_length` is BLIP's / CoCa's total length and is not read by
BLIP-2).  dtype: "f32s" (default for BLIP: fp32-grade split-fp16 GEMMs, token-identical to the fp32 reference), "bf16",
"f32"; None = the architecture's default.  CoCa: num_beam_groups (the model's beam groups, coca_model.py:218-219; None = one
group), tokenizer_dir (directory holding the CLIP BPE vocabulary - vocab.json / merges.txt / bpe_simple_vocab_16e6.txt.gz -
when it is not next to the checkpoint and open_clip is not installed); generation_type / top_k / top_p / temperature /
repetition_penalty (the reference model's other `generate` options, coca_model.py:205-224): accepted at their neutral values
("top_k" with top_k=1 = greedy, "beam_search"), rejected by name otherwise - sampling is not implemented."""


class Configuration:
    def __init__(self, arch_name=None, model_name=None, checkpoint_name=None, height=None, width=None, **extra):
        self.captioner = CaptionerField(arch_name=arch_name, model_name=model_name, checkpoint_name=checkpoint_name,
                                        height=height, width=width, **extra)


class CaptionerField:
    def __init__(self, arch_name=None, model_name=None, checkpoint_name=None, height=None, width=None,
                 num_beams=1, max_length=20, dtype=None, batch_size=8, device="cuda:0", image_size=None, streams=1,
                 early_exit_poll=None, max_new_tokens=None, num_beam_groups=None, tokenizer_dir=None,
                 generation_type=None, top_k=None, top_p=None, temperature=None, repetition_penalty=None,
                 load_in_8bit=None, load_in_4bit=None, torch_dtype=None, cross_cache=None, strict_range=False,
                 coalesce_rows=None, device_resize=None):
        self.arch_name = arch_name
        self.model_name = model_name
        self.checkpoint_name = checkpoint_name
        self.height = height
        self.width = width
        self.num_beams = num_beams
        self.max_length = max_length
        self.dtype = dtype
        self.batch_size = batch_size
        self.device = device
        self.image_size = image_size
        self.streams = streams
        self.early_exit_poll = early_exit_poll
        self.max_new_tokens = max_new_tokens
        self.num_beam_groups = num_beam_groups
        self.tokenizer_dir = tokenizer_dir
        # the reference model's other generation options (coca_model.py:205-224): None = not asked for; anything but the neutral
        # value is rejected by name when the captioner is built (captioner/generation_options.py)
        self.generation_type = generation_type
        self.top_k = top_k
        self.top_p = top_p
        self.temperature = temperature
        self.repetition_penalty = repetition_penalty
        # the reference's BLIP-2 load options (blip2.py:19-22): load_in_8bit = int8 Linear weights as bitsandbytes stores them + bf16
        # activations (models/blip2/blip2.py), load_in_4bit is rejected by name, torch_dtype only names the precision the
        # checkpoint is stored in
        self.load_in_8bit = load_in_8bit
        self.load_in_4bit = load_in_4bit
        self.torch_dtype = torch_dtype
        # "f32s" only: cross_cache "fp32" keeps fp32 rows in the cross-attention K/V cache instead of KV16 (None / "auto": KV16);
        # strict_range: a value that leaves the mode's range (clamped and counted by the library) raises instead of being logged
        self.cross_cache = cross_cache
        self.strict_range = strict_range
        # BLIP with streams > 1: the pool's dynamic batching merges the micro-batches of one generate_batch / caption_batch call
        # into passes of at most this many rows (same captions; None = 4 x batch_size up to 1024, 0 = off)
        self.coalesce_rows = coalesce_rows
        # PIL inputs: the processor's bicubic resize runs on the device, bit-exact with Pillow (None / True); False = host PIL
        self.device_resize = device_resize
