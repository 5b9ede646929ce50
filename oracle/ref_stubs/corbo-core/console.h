// ORACLE (test infrastructure only): corbo-core/console.h prints messages; here they are dropped.
#pragma once
#ifndef PRINT_ERROR_NAMED
#define PRINT_ERROR_NAMED(msg) do { } while (0)
#define PRINT_ERROR(msg) do { } while (0)
#define PRINT_ERROR_COND(cond, msg) do { } while (0)
#define PRINT_ERROR_COND_NAMED(cond, msg) do { } while (0)
#define PRINT_WARNING_COND_NAMED(cond, msg) do { } while (0)
#endif
