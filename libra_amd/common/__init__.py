from .registry import registry

__all__ = ["registry"]
