"""Plugin configuration objects - same fields as the reference's ``experimenting_env/captioner/utils/utils.py:2-12``
plus optional keys (defaults keep existing yamls working): num_beams, max_length, dtype, batch_size, device,
image_size (CoCa: open_clip's force_image_size), streams (engines / HIP streams the micro-batches of one call rotate over,
engine.EnginePool; 1 = one engine), early_exit_poll (look for "every caption finished" every n decode steps; None = 4)."""


class Configuration:
    def __init__(self, arch_name=None, model_name=None, checkpoint_name=None, height=None, width=None, **extra):
        self.captioner = CaptionerField(arch_name=arch_name, model_name=model_name, checkpoint_name=checkpoint_name,
                                        height=height, width=width, **extra)


class CaptionerField:
    def __init__(self, arch_name=None, model_name=None, checkpoint_name=None, height=None, width=None,
                 num_beams=1, max_length=20, dtype="bf16", batch_size=8, device="cuda:0", image_size=None, streams=1,
                 early_exit_poll=None):
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
