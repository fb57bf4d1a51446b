"""Drop-in shim for the reference's `utils` package (only `utils.util` is on the hot path)."""
