class EmptyClass:
    """Placeholder type: old R2L checkpoints pickle `args.trial` as utils.EmptyClass (reference utils/__init__.py)."""
    pass
