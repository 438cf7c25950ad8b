from .fast_plaid import FastPlaid  # noqa: F401

__all__ = ["FastPlaid"]
