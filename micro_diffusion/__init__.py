"""Import-path compatibility: the reference's package name resolves to the MI355X-native implementation, so
`from micro_diffusion.models.model import create_latent_diffusion` and the `_target_:` strings of the reference YAML
configs keep working unchanged."""
