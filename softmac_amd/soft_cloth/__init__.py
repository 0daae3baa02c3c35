"""Host-side mirror of the reference's `soft_cloth/` package for the part SURVEY 8 (row f4) puts in scope: the MPM substep variant
with von-Mises plasticity and a length scale, and its contact with one kinematically driven triangle-mesh sheet.  The cloth
dynamics themselves (DiffClothAI, closed source) are out of scope: `engine/cloth_simulator.py` holds a kinematic driver."""
