from .cli import main, main_causalbgm

__all__ = ["main", "main_causalbgm"]
