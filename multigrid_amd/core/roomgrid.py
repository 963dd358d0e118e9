"""`multigrid.core.roomgrid` of the reference, by name."""
from ..roomgrid import Room, RoomGrid, bfs, reject_next_to  # noqa: F401
