"""sparf_amd: MI355X-native NeRF volumetric renderer behind the SPARF `source.models.renderer` API."""
