"""``model.modules.adaptors`` import path (yaml ``multi_latent_adaptor.target``).  Implementation: camc2v_amd.adaptor."""
from camc2v_amd.adaptor import MultiLatentEpipolarAdaptor  # noqa: F401
