def create_model_and_transforms(*a, **k):
    raise RuntimeError("open_clip stub: text encoders are out of scope for the oracle")


def get_tokenizer(*a, **k):
    raise RuntimeError("open_clip stub")
