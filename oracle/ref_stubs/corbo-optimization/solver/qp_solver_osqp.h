// ORACLE (test infrastructure only): nothing of corbo::SolverOsqp is used by the code compiled here
#pragma once
