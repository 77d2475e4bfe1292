class AutoencoderKL:  # name only; never instantiated by the oracle
    pass
