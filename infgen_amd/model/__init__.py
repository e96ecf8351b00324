from .infgen import InfGen

__all__ = ['InfGen']
