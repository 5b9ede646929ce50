// ORACLE (test infrastructure only): nothing of corbo-communication is used by the sources compiled here.
#pragma once
