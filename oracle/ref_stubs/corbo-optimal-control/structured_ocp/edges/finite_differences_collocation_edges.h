// ORACLE (test infrastructure only): the collocation EDGES belong to createEdges (finite_differences_grid_se2.cpp), which is not compiled here.
#pragma once
