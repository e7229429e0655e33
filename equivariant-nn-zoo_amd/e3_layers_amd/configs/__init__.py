from .config_dict import ConfigDict
from . import config_energy, config_energy_force, config_diffusion, config_diffusion_CA, config_diffusion_backbone

__all__ = ["ConfigDict", "config_energy", "config_energy_force", "config_diffusion", "config_diffusion_CA",
           "config_diffusion_backbone"]
