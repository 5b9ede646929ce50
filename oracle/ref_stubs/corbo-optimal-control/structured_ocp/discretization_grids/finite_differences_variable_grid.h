// ORACLE (test infrastructure only): the corbo grid of this name is only included by the reference, not used
#pragma once
